"""Dev tool (GPU): how often does the recurrent cluster kernel raise its exchange-timeout word on small networks?"""
import sys
import time
sys.path.insert(0, '.')
import torch
import kraken_amd
from kraken_amd import _lib

specs = [('[1,9,0,1 Cr3,13,28 Mp2,2 Cr3,13,32 Mp2,2 Cr3,9,16 S1(1x0)1,3 Lbx8 O1c3]', 2, 260, [260, 131]),
         ('[1,16,0,1 Cr5,7,16 Mp2,2 Cr3,12,32 Mp2,2 Cr3,3,32 S1(1x0)1,3 Lbx8 O1c7]', 3, 301, [301, 300, 155]),
         ('[1,10,0,1 Cr1,16,32 Mp2,2 Cr3,11,32 Cr3,13,16 S1(1x0)1,3 Lbx8 Lbx8 O1c11]', 4, 517, [517, 516, 260, 31]),
         # the sizes lstm_ws.hip is the default for (H > 128), small and ragged batches
         ('[1,10,0,1 Cr1,16,32 Mp2,2 Cr3,11,32 Cr3,13,16 S1(1x0)1,3 Lbx136 Lbx160 O1c11]', 4, 517, [517, 516, 260, 31]),
         ('[1,10,0,1 Cr1,16,32 Mp2,2 Cr3,11,32 Cr3,13,16 S1(1x0)1,3 Lbx192 Lfx200 Lrx224 O1c11]', 7, 301, [301, 300, 155, 154, 40, 3, 1]),
         ('[1,10,0,1 Cr1,16,32 Mp2,2 Cr3,11,32 Cr3,13,16 S1(1x0)1,3 Lbx200 Lbx200 O1c11]', 40, 260, [260 - 5 * i for i in range(40)])]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
if len(sys.argv) > 2:
    specs = [specs[int(sys.argv[2])]]
for spec, n, w, lens in specs:
    torch.manual_seed(0)
    m = kraken_amd.TorchVGSLModel(vgsl=spec)
    m.nn.set_precision('bf16x3')
    m.to('cuda')
    x = torch.rand(n, 1, int(spec.split(',')[1]), w).cuda()
    fails, t0 = 0, time.time()
    for i in range(reps):
        try:
            m.nn(x, torch.tensor(lens))
        except _lib.KrakenAmdError as e:
            fails += 1
            if fails <= 2:
                print('   ', str(e)[60:150])
    print(f'{spec[-28:]} N={n} W={w}: {fails} timeouts in {reps} forwards, {time.time() - t0:.1f} s', flush=True)

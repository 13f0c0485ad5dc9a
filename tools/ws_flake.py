"""Dev tool (GPU): how often does the recurrent cluster kernel raise its exchange-timeout word on small networks?
    python tools/ws_flake.py [forwards] [spec index]         one forward at a time (nn(x): synchronises, reads the status word)
    python tools/ws_flake.py [batches] [spec index] --slots  the same networks through the pipelined engine, three batches in flight on
                                                            three streams (clusters of different launches compete for the CUs; with
                                                            KRK_LSTM_V=3 the cluster kernel also takes the narrow layers)"""
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
         ('[1,10,0,1 Cr1,16,32 Mp2,2 Cr3,11,32 Cr3,13,16 S1(1x0)1,3 Lbx200 Lbx200 O1c11]', 40, 260, [260 - 5 * i for i in range(40)]),
         # round 6: hidden sizes that are not a multiple of 8 on the cluster kernel (directions written Hp units wide), eight K blocks (256)
         ('[1,10,0,1 Cr1,16,32 Mp2,2 Cr3,11,32 Cr3,13,16 S1(1x0)1,3 Lbx100 Lbx150 Lfx75 O1c11]', 7, 301, [301, 300, 155, 154, 40, 3, 1]),
         ('[1,10,0,1 Cr1,16,32 Mp2,2 Cr3,11,32 Cr3,13,16 S1(1x0)1,3 Lbx256 Lbx250 O1c11]', 20, 260, [260 - 11 * i for i in range(20)])]
args = [a for a in sys.argv[1:] if not a.startswith('--')]
reps = int(args[0]) if args else 200
if len(args) > 1:
    specs = [specs[int(args[1])]]
if '--slots' in sys.argv:
    import logging
    import numpy as np
    from kraken_amd.engine import RecognitionEngine

    class _Count(logging.Handler):            # the engine re-runs a timed-out batch once on the streaming kernel and says so
        n = 0

        def emit(self, record):
            if 'streaming recurrent kernel' in record.getMessage():
                _Count.n += 1
    logging.getLogger('kraken_amd.engine').addHandler(_Count())
    for spec, n, w, lens in specs:
        torch.manual_seed(0)
        m = kraken_amd.TorchVGSLModel(vgsl=spec, codec={chr(0x61 + i): [i + 1] for i in range(int(spec.rstrip(']').split('O1c')[1]) - 1)})
        m.nn.set_precision('bf16x3')
        m.to('cuda')
        eng = RecognitionEngine(m, device=0, max_batch=max(n, 8), max_width=w, slots=3)
        xs = [torch.rand(n, 1, int(spec.split(',')[1]), w).cuda() for _ in range(3)]
        la = np.asarray(lens, np.int32)
        first, fails, t0 = None, 0, time.time()
        for i in range(reps):
            if eng.free_slots() == 0:
                try:
                    eng.collect()
                except _lib.KrakenAmdError:
                    fails += 1
            eng.submit(xs[i % 3], la)
        while eng.free_slots() < 3:
            try:
                eng.collect()
            except _lib.KrakenAmdError:
                fails += 1
        eng.close()
        print(f'{spec[-28:]} N={n} W={w}: {_Count.n} exchange timeouts (retried), {fails} failed batches in {reps} (3 in flight), '
              f'{time.time() - t0:.1f} s', flush=True)
        _Count.n = 0
    sys.exit(0)
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

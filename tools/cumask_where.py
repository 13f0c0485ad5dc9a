import ctypes as C, collections, sys
import torch
sys.path.insert(0, '.')
hip = C.CDLL('libamdhip64.so')
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
lib = C.CDLL('kraken_amd/_exp_cumask.so')
lib.where_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong]
torch.cuda.init()
def masked(bits):
    words = (C.c_uint32 * 8)(*[0] * 8)
    for b in bits: words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p(); assert hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words) == 0
    return s
def run(name, bits):
    s = masked(list(bits)) if bits is not None else C.c_void_p(torch.cuda.current_stream().cuda_stream)
    n = 2048
    out = torch.zeros(2 * n, dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    assert lib.where_launch(s, out.data_ptr(), n, 40000) == 0
    torch.cuda.synchronize()
    o = out.cpu().numpy().astype('uint32').reshape(n, 2)
    hw, xcc = o[:, 0], o[:, 1] & 0xf
    cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 0x7
    ids = set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    per_xcc = collections.Counter(x for x, *_ in ids)
    print(f'{name:18s} distinct CUs {len(ids):4d}  per XCC {dict(sorted(per_xcc.items()))}')
    return ids
all_ids = run('default', None)
run('all 256', range(256)); run('first 128', range(128)); run('first 64', range(64)); run('even bits', range(0, 256, 2))
run('first 160', range(160)); run('last 96', range(160, 256)); run('bits 0..7', range(8)); run('bits 0,8,16..', range(0, 256, 8))

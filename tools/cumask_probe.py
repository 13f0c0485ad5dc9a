"""Does hipExtStreamCreateWithCUMask restrict kernels to a CU subset on this box, and how are mask bits mapped?
Times a fixed torch matmul on streams with different masks."""
import ctypes as C
import torch

hip = C.CDLL('libamdhip64.so')
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = C.c_int


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[0] * 8)
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


torch.cuda.init()
a = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)
b = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)


def t(stream):
    with torch.cuda.stream(stream):
        for _ in range(3):
            a @ b
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record(stream)
        for _ in range(10):
            a @ b
        e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / 10


print('default stream      %.3f ms' % t(torch.cuda.current_stream()))
for name, bits in [('all 256', range(256)), ('first 128', range(128)), ('first 64', range(64)), ('even bits (128)', range(0, 256, 2)),
                   ('first 160', range(160)), ('last 96', range(160, 256))]:
    print('%-18s %.3f ms' % (name, t(masked_stream(list(bits)))))

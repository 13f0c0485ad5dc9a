"""
ctypes binding of ``libkraken_amd.so`` (C ABI declared in ``include/kraken_amd.h``).

There is deliberately NO fallback: if the shared library is missing or a HIP
device is absent, the product path raises.  Only ``oracle/`` (test
infrastructure) computes this path on a CPU.
"""
import ctypes as C
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
# KRAKEN_AMD_LIB: alternative build of the same library (kernel experiments); still a HIP build, never a fallback
LIB_PATH = os.environ.get('KRAKEN_AMD_LIB') or os.path.join(HERE, 'libkraken_amd.so')

KRK_OK = 0
KRK_E_INVALID, KRK_E_HIP, KRK_E_NOMEM, KRK_E_UNSUPPORTED = -1, -2, -3, -4

OP_CONV, OP_MAXPOOL, OP_GROUPNORM, OP_RESHAPE_HC, OP_LSTM, OP_LINEAR = 1, 2, 3, 4, 5, 6
OP_PAR_BEGIN, OP_PAR_NEXT, OP_PAR_END, OP_ADD, OP_CONVT, OP_RESHAPE = 7, 8, 9, 10, 11, 12
ACT_LINEAR, ACT_RELU, ACT_TANH, ACT_LEAKY, ACT_SIGMOID, ACT_SOFTMAX = 0, 1, 2, 3, 4, 5
DIR_FWD, DIR_REV, DIR_BIDI = 0, 1, 2
PREC_F32, PREC_BF16, PREC_BF16X3 = 0, 1, 2

# every symbol include/kraken_amd.h declares (checked by the CPU test-suite)
EXPORTS = ['krk_abi_version', 'krk_last_error', 'krk_device_count', 'krk_plan_create', 'krk_plan_destroy',
           'krk_plan_out_shape', 'krk_plan_olens', 'krk_forward', 'krk_greedy_decode', 'krk_recognize',
           'krk_plan_workspace_bytes', 'krk_plan_set_profiling', 'krk_plan_layer_ms', 'krk_plan_layer_name',
           'krk_plan_layer_flops', 'krk_plan_num_steps', 'krk_plan_front_event', 'krk_plan_wait_front',
           'krk_plan_status', 'krk_prep_lines', 'krk_prep_crops', 'krk_upsample_sigmoid', 'krk_dewarp_measure', 'krk_dewarp_apply',
           'krk_prep_lines_fmt', 'krk_dewarp_measure_page', 'krk_dewarp_apply_page', 'krk_plan_has_exchange',
           'krk_plan_set_recurrence', 'krk_plan_out_dims', 'krk_plan_olens_w', 'krk_plan_get_recurrence', 'krk_plan_clone']


class KrkLayer(C.Structure):
    _fields_ = [('op', C.c_int), ('cout', C.c_int),
                ('kh', C.c_int), ('kw', C.c_int), ('sh', C.c_int), ('sw', C.c_int), ('dh', C.c_int), ('dw', C.c_int),
                ('act', C.c_int), ('direction', C.c_int),
                ('w', C.c_void_p * 8)]


class KrkDecodeOut(C.Structure):
    _fields_ = [('labels', C.c_void_p), ('starts', C.c_void_p), ('ends', C.c_void_p), ('confs', C.c_void_p),
                ('counts', C.c_void_p), ('t_stride', C.c_int)]


class KrakenAmdError(RuntimeError):
    """Raised when an entry point of libkraken_amd.so reports a failure."""

    def __init__(self, code, msg):
        super().__init__(f'libkraken_amd error {code}: {msg}')
        self.code = code


def is_exchange_timeout(e: Exception) -> bool:
    """The status word of a plan (krk_plan_status): a recurrent cluster kernel gave up waiting for its peers."""
    return isinstance(e, KrakenAmdError) and 'timed out waiting for its peers' in str(e)


RECURRENCE_AUTO, RECURRENCE_STREAMING = 0, 1
ABI_VERSION = 3          # include/kraken_amd.h: KRK_ABI_VERSION


class streaming_recurrence:
    """Context: forward calls of ``plan`` (a handle) made inside use the streaming recurrent kernel (lstm_x3.hip, no inter-workgroup
    exchange) instead of the cluster kernel -- the one retry after an exchange timeout.  The switch is a field of THAT plan
    (krk_plan_set_recurrence): other plans, engine slots and threads are not touched, and nothing is written to the process
    environment (KRK_LSTM_V stays what it is: a debugging probe)."""

    def __init__(self, plan_handle):
        self.handle, self.before = plan_handle, RECURRENCE_AUTO

    def __enter__(self):
        lib = load()
        before = lib.krk_plan_get_recurrence(self.handle)
        if before < 0:
            check(before)
        self.before = before                     # a plan its owner had set to STREAMING stays there after the retry
        check(lib.krk_plan_set_recurrence(self.handle, RECURRENCE_STREAMING))

    def __exit__(self, *exc):
        check(load().krk_plan_set_recurrence(self.handle, self.before))


def checked_run(plan_handle, run, wait, log=None):
    """
    The one policy for a batch whose recurrent cluster kernel gave up waiting for its peers, used by ``nn(x)`` and
    ``nn.recognize`` (``RecognitionEngine.collect`` applies the same rule to a batch that is ALREADY enqueued on a slot's stream --
    it re-launches through its own slot bookkeeping, with the same ``streaming_recurrence`` switch): ``run()`` enqueues the batch, ``wait()`` blocks until it has completed;
    the plan's status word is read then, and an exchange timeout re-runs THIS batch once on the streaming kernel (a warning, not
    a lost page).  Anything else -- and a second failure -- raises.
    """
    lib = load()
    run()
    wait()
    try:
        check(lib.krk_plan_status(plan_handle))
    except KrakenAmdError as e:
        if not is_exchange_timeout(e):
            raise
        if log is not None:
            log.warning(f'{e}; running this batch again on the streaming recurrent kernel')
        with streaming_recurrence(plan_handle):
            run()
            wait()
        check(lib.krk_plan_status(plan_handle))


_lib = None
_lock = threading.Lock()


def load():
    """Loads the shared library (once). Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(f'{LIB_PATH} is missing: build the HIP extension first '
                              '(python -m kraken_amd.build, or __graft_entry__.build()). '
                              'kraken_amd has no CPU fallback.')
        lib = C.CDLL(LIB_PATH)
        vp, i32, f32, lng = C.c_void_p, C.c_int, C.c_float, C.c_long
        lib.krk_abi_version.restype = i32
        # the version FIRST: a stale library must say "rebuild", not die of an AttributeError on an export it lacks
        if lib.krk_abi_version() != ABI_VERSION:
            raise ImportError(f'{LIB_PATH}: ABI version {lib.krk_abi_version()}, this package needs {ABI_VERSION}; '
                              'rebuild the extension (python -m kraken_amd.build)')
        lib.krk_last_error.restype = C.c_char_p
        lib.krk_device_count.restype = i32
        lib.krk_plan_create.argtypes = [C.POINTER(KrkLayer), i32, i32, i32, i32, i32, C.POINTER(vp)]
        lib.krk_plan_create.restype = i32
        lib.krk_plan_clone.argtypes = [vp, C.POINTER(vp)]
        lib.krk_plan_clone.restype = i32
        lib.krk_plan_destroy.argtypes = [vp]
        lib.krk_plan_destroy.restype = None
        lib.krk_plan_out_shape.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
        lib.krk_plan_out_shape.restype = i32
        lib.krk_plan_olens.argtypes = [vp, vp, i32, vp]
        lib.krk_plan_olens.restype = i32
        lib.krk_plan_out_dims.argtypes = [vp, i32, i32] + [C.POINTER(i32)] * 4
        lib.krk_plan_out_dims.restype = i32
        lib.krk_plan_olens_w.argtypes = [vp, vp, i32, i32, vp]
        lib.krk_plan_olens_w.restype = i32
        lib.krk_forward.argtypes = [vp, vp, vp, i32, i32, vp, vp]
        lib.krk_forward.restype = i32
        lib.krk_greedy_decode.argtypes = [vp, lng, lng, lng, i32, i32, i32, vp, i32, f32, vp, vp,
                                          C.POINTER(KrkDecodeOut)]
        lib.krk_greedy_decode.restype = i32
        lib.krk_recognize.argtypes = [vp, vp, vp, i32, i32, f32, vp, vp, vp, vp, C.POINTER(KrkDecodeOut)]
        lib.krk_recognize.restype = i32
        lib.krk_plan_workspace_bytes.argtypes = [vp]
        lib.krk_plan_workspace_bytes.restype = lng
        lib.krk_plan_set_profiling.argtypes = [vp, i32]
        lib.krk_plan_set_profiling.restype = i32
        lib.krk_plan_layer_ms.argtypes = [vp, vp, i32]
        lib.krk_plan_layer_ms.restype = i32
        lib.krk_plan_layer_name.argtypes = [vp, i32]
        lib.krk_plan_layer_name.restype = C.c_char_p
        lib.krk_plan_layer_flops.argtypes = [vp, i32]
        lib.krk_plan_layer_flops.restype = C.c_double
        lib.krk_plan_num_steps.argtypes = [vp]
        lib.krk_plan_num_steps.restype = i32
        lib.krk_plan_front_event.argtypes = [vp]
        lib.krk_plan_front_event.restype = vp
        lib.krk_plan_wait_front.argtypes = [vp, vp]
        lib.krk_plan_wait_front.restype = i32
        lib.krk_plan_status.argtypes = [vp]
        lib.krk_plan_status.restype = i32
        lib.krk_plan_has_exchange.argtypes = [vp]
        lib.krk_plan_has_exchange.restype = i32
        lib.krk_plan_set_recurrence.argtypes = [vp, i32]
        lib.krk_plan_set_recurrence.restype = i32
        lib.krk_plan_get_recurrence.argtypes = [vp]
        lib.krk_plan_get_recurrence.restype = i32
        lib.krk_prep_lines.argtypes = [vp, i32, i32, i32, vp, i32, i32, i32, i32, i32, vp, vp, vp]
        lib.krk_prep_lines.restype = i32
        lib.krk_prep_crops.argtypes = [vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, vp]
        lib.krk_prep_crops.restype = i32
        lib.krk_upsample_sigmoid.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp]
        lib.krk_upsample_sigmoid.restype = i32
        lib.krk_dewarp_measure.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp]
        lib.krk_dewarp_measure.restype = i32
        lib.krk_dewarp_apply.argtypes = [vp, vp, i32, i32, vp, vp, i32, i32, i32, vp, vp, vp]
        lib.krk_dewarp_apply.restype = i32
        i64 = C.c_long
        lib.krk_prep_lines_fmt.argtypes = [vp, i32, i32, i64, i32, i32, vp, i32, i32, i32, i32, i32, vp, vp, vp]
        lib.krk_prep_lines_fmt.restype = i32
        lib.krk_dewarp_measure_page.argtypes = [vp, i64, i32, vp, i32, i32, i32, vp, vp, vp, vp, vp]
        lib.krk_dewarp_measure_page.restype = i32
        lib.krk_dewarp_apply_page.argtypes = [vp, i64, i32, vp, i32, i32, vp, vp, i32, i32, i32, vp, vp, vp]
        lib.krk_dewarp_apply_page.restype = i32
        _lib = lib
    return _lib


def check(rc: int):
    if rc != KRK_OK:
        msg = load().krk_last_error()
        raise KrakenAmdError(rc, msg.decode('utf-8', 'replace') if msg else '')


def device_count() -> int:
    return int(load().krk_device_count())


def require_gpu():
    """The product path needs a HIP device; fail loudly otherwise."""
    if device_count() <= 0:
        raise KrakenAmdError(KRK_E_HIP, 'no HIP device visible: the kraken_amd recognition path runs only on '
                                        'MI355X-class GPUs and has no CPU fallback')

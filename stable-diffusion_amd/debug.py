"""Debug aids of libsdmi that have no counterpart in the reference.

fp16 range guard: every MFMA operand of the HIP path is fp16 (the reference runs the same tensors in fp16 under
`torch.autocast`, scripts/txt2img.py:283).  Synthetic weights keep activations O(1); a real checkpoint has outlier
channels.  `range_check(True)` makes the library scan every fp16 activation buffer right after the launch that wrote it
(GroupNorm / LayerNorm outputs, q / k / v^T, GEGLU, attention output, fp16 copies of the residual stream);
`range_report()` returns the totals and the first offending kernel class.  The scan synchronises the stream after each
launch -- a debugging mode (also SDMI_CHECK_RANGE=1 in the environment), never on in a timed run.

What to do when it trips: the overflow is in ONE operand of ONE GEMM (the report names it).  The residual stream, all
statistics and all accumulators are fp32, so nothing upstream is damaged; the remedy is local to that operand (scale it by
2^-k where it is produced and by 2^k in the consuming GEMM's fp32 epilogue -- exact for powers of two).  The guard, not a
blanket bf16 fallback, is what ships: bf16 operands cannot meet the 1e-3 parity bar (1.9e-2).
"""
import ctypes as C
import json

from . import _lib


def range_check(enable=True):
    """Enable / disable the fp16 range guard; either way the counters are cleared."""
    _lib.check(_lib.load().sdmi_range_check(1 if enable else 0))


def range_report():
    """{'over_6e4': int, 'nonfinite': int, 'max_abs': float, 'first': str} since the last range_check() call."""
    buf = C.create_string_buffer(1024)
    _lib.check(_lib.load().sdmi_range_report(buf, len(buf)))
    return json.loads(buf.value.decode())

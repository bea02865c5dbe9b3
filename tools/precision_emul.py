"""python tools/precision_emul.py <golden case> [--only | --drop ... | --per-layer]: which operand roundings carry the eps error
of a golden case (CPU emulation of the fp16-operand design on the oracle; the implementation lives in oracle/fp16_floor.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.fp16_floor import main  # noqa: E402

if __name__ == '__main__':
    main()

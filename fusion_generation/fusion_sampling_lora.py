#!/usr/bin/env python3
"""Drop-in for `python fusion_generation/fusion_sampling_lora.py ...` (rank-4 LoRA concept weights, `--t_stop`
window of fusion_sampling_lora.py:324,378,476-492).  Same flags as fusion_sampling.py plus --t_stop."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fusion_sampling as _fs  # noqa: E402

_fs.LORA = True

if __name__ == '__main__':
    _fs.main()

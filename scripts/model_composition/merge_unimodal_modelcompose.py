#!/usr/bin/env python
# Usage: python merge_unimodal_modelcompose.py ckpt1 ckpt2 ... -o merged --strategy online-merge-reset-default-video=0.333,...
# Same CLI as the reference script of the same path; implementation in modelcompose_amd/compose.py.
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from modelcompose_amd.compose import main  # noqa: E402

if __name__ == "__main__":
    main()

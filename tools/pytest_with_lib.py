#!/usr/bin/env python3
"""usage: YAK_LIB=path/to/libyakhip.so python tools/pytest_with_lib.py <pytest arguments>
Runs the Python tests against another build of the library (e.g. one whose host code is compiled with
-Xarch_host -fsanitize=undefined): yak_amd.capi loads the in-tree library by a fixed path, this points it elsewhere first."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytest  # noqa: E402

from yak_amd import capi  # noqa: E402

capi.LIB_PATH = os.path.abspath(os.environ["YAK_LIB"])
print("library under test:", capi.LIB_PATH, flush=True)
sys.exit(pytest.main(sys.argv[1:]))

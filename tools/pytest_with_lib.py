"""pytest against a VARIANT of the library (same ABI): python tools/pytest_with_lib.py qfa_amd/libqfa_<variant>.so <pytest args ...>
(the __main__ guard matters: tools/oracle_pool.py starts spawn-context workers, which import the main module again)"""
import os, sys

if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from qfa_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
    import pytest
    raise SystemExit(pytest.main(sys.argv[2:]))

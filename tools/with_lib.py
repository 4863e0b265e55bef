"""Run a script (default: bench.py) against a VARIANT of the library, for same-box A/B timing and the stamp builds:
    python tools/with_lib.py qfa_amd/libqfa_<variant>.so [script.py] [args ...]
The product (qfa_amd/_lib.py) reads no environment variable; this tool sets _lib.LIB_PATH before the first lib() call.  The
variant must have the shipped ABI (qfa_abi_version) and exports: _lib refuses anything else."""
import os, runpy, sys

if __name__ == "__main__":          # (guarded: a script that starts spawn-context workers imports this module again in each of them)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from qfa_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
    rest = sys.argv[2:]
    script = os.path.join(root, "bench.py")
    if rest and rest[0].endswith(".py"):
        script, rest = rest[0], rest[1:]
    sys.argv = [script] + rest
    runpy.run_path(script, run_name="__main__")

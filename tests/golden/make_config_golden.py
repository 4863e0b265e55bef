"""Build-container script: extract the reference's configuration schema into tests/golden/g12_config.json.

yacs is not installed here, so the reference's QFA/config.py cannot be imported; its content that matters for
parity is plain data, read with ``ast`` from the source text WHERE IT LIES (/root/reference, never copied):
  * defaults  -- every ``_C.A.B = <literal>`` assignment of QFA/config.py:15-63
  * arg_keys  -- every ``if _check_args('x'): config.A.B = args.x`` of QFA/config.py:92-139
  * flags     -- every ``parser.add_argument("--x", type=T, ...)`` of main.py:16-42 (name, type, nargs)
The JSON holds those three tables only (data, no source text).  Run:  python tests/golden/make_config_golden.py
"""
import ast
import json
import os
import sys

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "g12_config.json")


def dotted(node):
    parts = []
    while isinstance(node, ast.Attribute):
        parts.append(node.attr)
        node = node.value
    if isinstance(node, ast.Name):
        parts.append(node.id)
    return list(reversed(parts))


def main():
    cfg_tree = ast.parse(open(os.path.join(REF, "QFA", "config.py")).read())
    defaults, arg_keys = {}, {}
    for node in ast.walk(cfg_tree):
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Attribute):
            path = dotted(node.targets[0])
            if path[0] == "_C":
                try:
                    defaults[".".join(path[1:])] = ast.literal_eval(node.value)
                except ValueError:
                    pass                                    # _C.DATA = CN() etc.: a sub-node, not a value
        if isinstance(node, ast.If) and isinstance(node.test, ast.Call) and getattr(node.test.func, "id", "") == "_check_args":
            flag = ast.literal_eval(node.test.args[0])
            tgt = node.body[0].targets[0]
            path = dotted(tgt)
            assert path[0] == "config"
            arg_keys[flag] = ".".join(path[1:])
    main_tree = ast.parse(open(os.path.join(REF, "main.py")).read())
    flags = {}
    for node in ast.walk(main_tree):
        if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument":
            name = ast.literal_eval(node.args[0]).lstrip("-")
            kw = {k.arg: k.value for k in node.keywords}
            flags[name] = {"type": kw["type"].id if "type" in kw else None,
                           "nargs": ast.literal_eval(kw["nargs"]) if "nargs" in kw else None}
    json.dump({"source": "ZechangSun/QFA QFA/config.py:15-63,92-139 and main.py:16-42 (parsed with ast)",
               "defaults": defaults, "arg_keys": arg_keys, "flags": flags}, open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote", OUT, len(defaults), "defaults,", len(arg_keys), "arg keys,", len(flags), "flags")


if __name__ == "__main__":
    sys.exit(main())

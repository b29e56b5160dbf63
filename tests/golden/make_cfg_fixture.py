#!/usr/bin/env python3
"""Parses the reference's dynamic_reconfigure description (cfg/LidarFilters.cfg) into
tests/golden/lidar_filters_cfg.json, the fixture tests/test_param_table.py compares the library's
parameter table with on machines where /root/reference is not mounted.
    python tests/golden/make_cfg_fixture.py [/root/reference/cfg/LidarFilters.cfg]"""
import ast
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT = "/root/reference/cfg/LidarFilters.cfg"


def parse_cfg(path):
    """One dict per gen.add(...): name, type, default, min, max (None where the cfg gives none:
    dynamic_reconfigure then uses the type's full range), line; enum constants of edit_method."""
    src = open(path).read()
    tree = ast.parse(src)
    enums = {}
    rows = []
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and isinstance(node.value, ast.Call) and getattr(node.value.func, "attr", "") == "enum":
            consts = []
            for c in node.value.args[0].elts:   # gen.const(name, type, value, descr)
                consts.append((ast.literal_eval(c.args[0]), ast.literal_eval(c.args[2])))
            enums[node.targets[0].id] = consts
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add":
            a = node.args
            name = ast.literal_eval(a[0])
            typ = a[1].id            # str_t / bool_t / int_t / double_t
            default = ast.literal_eval(a[4])
            lo = ast.literal_eval(a[5]) if len(a) > 5 else None
            hi = ast.literal_eval(a[6]) if len(a) > 6 else None
            enum = None
            for kw in node.keywords:
                if kw.arg == "edit_method":
                    enum = enums[kw.value.id]
            rows.append(dict(name=name, type=typ, default=default, min=lo, max=hi, line=node.lineno, enum=enum))
    rows.sort(key=lambda r: r["line"])
    return rows


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else DEFAULT
    rows = parse_cfg(path)
    out = os.path.join(HERE, "lidar_filters_cfg.json")
    json.dump(rows, open(out, "w"), indent=1)
    print("%d parameters -> %s" % (len(rows), out))

#!/usr/bin/env python
"""`vit` command line (see vit.cpp_amd/cli.py): python vit_cli.py -m model.gguf -i image.jpg -k 5"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _pkg  # noqa: E402

pkg = _pkg.load()
from vitcpp_amd import cli  # noqa: E402

if __name__ == "__main__":
    raise SystemExit(cli.main())

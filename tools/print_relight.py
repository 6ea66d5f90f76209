"""The relight entry of a bench.py JSON line: python tools/print_relight.py <file>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["roofline"].get("relight"))

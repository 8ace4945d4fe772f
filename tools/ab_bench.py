#!/usr/bin/env python3
"""Same-box A/B of kernel variants: runs bench.py alternately with each library (SPX_LIB) and prints the
throughputs -- boxes of the pool differ by 3-4 %, so two variants are only comparable inside one gpurun call.
  python tools/ab_bench.py [--rounds 3] [--platform hifi|ont] name=path/to/libspx.so ...   (name "cur" = the in-tree build)"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--platform", default="hifi")
    ap.add_argument("--mode", default="kernel", choices=["kernel", "pipe"], help="kernel: --kernel-only replay; pipe: the whole pipelined step")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    res = {}
    for _ in range(a.rounds):
        for v in a.variants:
            name, _, path = v.partition("=")
            env = dict(os.environ)
            env.pop("SPX_LIB", None)
            if path:
                env["SPX_LIB"] = os.path.abspath(path)
            extra = ["--kernel-only"] if a.mode == "kernel" else ["--no-host-leg"]
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "3", "--no-cpu-baseline",
                                  "--verify", "0", "--platform", a.platform] + extra, env=env, capture_output=True, text=True).stdout
            d = json.loads(out.strip().splitlines()[-1])
            res.setdefault(name, []).append(d["value"])
    for name, vals in res.items():
        print(f"{name:12s} mean {sum(vals) / len(vals):10.0f}  runs {[round(x) for x in vals]}")


if __name__ == "__main__":
    main()

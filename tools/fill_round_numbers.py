"""Fills the R6_* placeholders of DESIGN.md / README.md / BASELINE.md from the committed bench records under profiles/ (one-off per round:
the documents quote exactly what profiles/r06_bench_n1.json and its siblings say).  python tools/fill_round_numbers.py [--check]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last(path):
    return json.loads(open(os.path.join(ROOT, path)).read().strip().splitlines()[-1])


def main():
    d = last("profiles/r06_bench_n1.json")
    r = d["roofline"]
    seg = d.get("segmenter_step") or last("profiles/r06_bench_segmenter_n1.json")
    b32 = last("profiles/r06_bench_bf16_B32_n1.json")
    v = {
        "R6_JOINT_MS": "%.1f" % d["ms_per_step"],
        "R6_JOINT": "%.1f" % d["value"],
        "R6_ALGF": "%.2f" % d["step_algorithmic"]["frac_of_mfma_peak"],
        "R6_ALG": "%.0f" % d["step_algorithmic"]["achieved"],
        "R6_FP32": "%.1f" % d["fp32_mfma_step"]["value"],
        "R6_US": "%.1f" % (1e3 * r["avg_launch_ms"]),
        "R6_TF": "%.0f" % r["achieved"],
        "R6_FRAC": "%.3f" % r["frac"],
        "R6_EQ": "%.0f" % r.get("achieved_fp32_equivalent", r["achieved"] / 6.0),
        "R6_SEG_MS": "%.1f" % seg["ms_per_step"],
        "R6_SEG_ALG": "%.0f" % (248.95 * 16 / seg["ms_per_step"]),
        "R6_SEG": "%.0f" % seg["value"],
        "R6_BF16_32": "%.0f" % b32["value"],
        "R6_BF16": "%.0f" % d["bf16_step"]["value"],
        "R6_CPU": "%.3f" % d["cpu_baseline"]["value"],
    }
    check = "--check" in sys.argv
    for name in ("DESIGN.md", "README.md", "BASELINE.md"):
        p = os.path.join(ROOT, name)
        s = open(p).read()
        n = 0
        for k in sorted(v, key=len, reverse=True):           # longest first: R6_JOINT_MS before R6_JOINT
            n += s.count(k)
            s = s.replace(k, v[k])
        print(name, n, "placeholders")
        if not check:
            open(p, "w").write(s)
    print(v)


if __name__ == "__main__":
    main()

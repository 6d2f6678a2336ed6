"""Per kernel symbol: per-launch averages of whatever counters the given rocprofv3 --pmc csv passes hold (no derived metrics)."""
import sys

from pmc_summary import avg, read_pass


def main():
    table = {}
    for folder in sys.argv[1:]:
        try:
            vals, dur = read_pass(folder)
        except SystemExit as e:
            print("# %s: %s" % (folder, e))
            continue
        for k, cs in vals.items():
            row = table.setdefault(k, {})
            for c, xs in cs.items():
                row[c] = avg(xs)
            row.setdefault("launches", len(dur[k]))
            row["dur_us_" + folder.rstrip("/").split("/")[-1]] = avg(list(dur[k].values())) / 1e3
    for k in sorted(table, key=lambda k: -table[k].get("SQ_WAVE_CYCLES", 0)):
        if "conv" not in k and "splitk" not in k and "bn_" not in k and "colreduce" not in k:
            continue
        print(k)
        for c, v in sorted(table[k].items()):
            print("    %-28s %.6g" % (c, v))


if __name__ == "__main__":
    sys.path.insert(0, __file__.rsplit("/", 1)[0])
    main()

"""CPU: the evidence tooling (tools/) on synthetic inputs — a broken summary script costs GPU minutes to discover."""
import importlib.util
import os
import sqlite3

from conftest import ROOT


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_rocpd_summary_isolates_the_timed_region(tmp_path, capsys):
    """rocpd_summary --last-ms: only the dispatches that start within the last X ms of the trace (bench.py's timed region; warm-up and
    the BN calibration forwards of the joint workload come before it)"""
    db = str(tmp_path / "trace.db")
    con = sqlite3.connect(db)
    con.execute("create table kernels (name text, start integer, end integer, duration integer)")
    rows = []
    t = 1_000_000_000
    for i in range(40):                                   # calibration: 40 forwards of 1 ms, long before the timed region
        rows.append(("void (anonymous namespace)::conv_taps_kernel<128, 128>(pnpconv::ConvArgs)", t, t + 1_000_000, 1_000_000))
        t += 1_500_000
    t += 50_000_000
    for i in range(10):                                   # timed region: 10 x (2 ms conv + 0.5 ms bn) back to back = 25 ms
        rows.append(("void (anonymous namespace)::conv_taps_kernel<128, 128>(pnpconv::ConvArgs)", t, t + 2_000_000, 2_000_000))
        t += 2_000_000
        rows.append(("bn_apply_kernel<true>(BnApplyArgs)", t, t + 500_000, 500_000))
        t += 500_000
    con.executemany("insert into kernels values (?, ?, ?, ?)", rows)
    con.commit()
    con.close()
    mod = _load("rocpd_summary")
    out = str(tmp_path / "all.txt")
    mod.main(db, out)
    text = open(out).read()
    assert "over 60 dispatches" in text and "conv_taps_kernel<128, 128>" in text and "(anonymous namespace)" not in text
    out2 = str(tmp_path / "timed.txt")
    mod.main(db, out2, last_ms=25.0)
    text2 = open(out2).read()
    assert "over 20 dispatches" in text2 and "total kernel time 25.000 ms" in text2
    conv = [l for l in text2.splitlines() if l.startswith("conv_taps_kernel")][0].split()
    assert conv[-6:] == ["10", "20.000", "2000.0", "2000.0", "2000.0", "80.00"]
    capsys.readouterr()


def test_pmc_summary_shortens_kernel_names_like_the_kernel_tables():
    mod = _load("pmc_summary")
    assert mod.clean("void (anonymous namespace)::conv_taps_kernel<128, 128, 2, 2, 0, 3, 3>(pnpconv::ConvArgs)") == \
        "conv_taps_kernel<128, 128, 2, 2, 0, 3, 3>"


def test_shell_tooling_parses_and_sanitizer_targets_exist():
    """tools/*.sh are only ever executed on a GPU box, minutes into a paid call: a syntax error there costs the call (round 3 lost one
    collection to a merge limit, not to syntax — keep it that way).  The Makefile's sanitizer targets are named in profiles/README.md."""
    import glob
    import subprocess
    scripts = sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh")) + glob.glob(os.path.join(ROOT, "tools", "experiments", "*.sh")))
    assert os.path.join(ROOT, "tools", "collect_round.sh") in scripts
    for s in scripts:
        r = subprocess.run(["bash", "-n", s], capture_output=True, text=True)
        assert r.returncode == 0, (s, r.stderr)
    mk = open(os.path.join(ROOT, "medical-cross-modality-domain-adaptation_amd", "csrc", "Makefile")).read()
    for target in ("asan:", "ubsan:", "variant:"):
        assert "\n" + target in mk, target
    body = open(os.path.join(ROOT, "tools", "collect_round.sh")).read()
    assert "rm -rf $O/pmc_fetch" in body          # raw PMC CSVs removed before gpurun merges gpurun_out/ back (64 MiB limit)

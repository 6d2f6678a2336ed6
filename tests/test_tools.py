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


def test_bench_last_line_is_compact_whatever_the_kernel_count():
    """BENCH_r03.json arrived `parsed: null`: bench.py printed one ~20 KB line (a 42-entry per-symbol table) and the driver keeps only a
    tail of stdout.  The record the driver parses is now built by bench.compact_record: < 2000 characters for ANY number of kernel
    symbols, round-trips through json, and carries every contract field plus roofline / cpu_baseline."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    rows = [{"name": "conv_taps_kernel<128, %d, 2, 2, %d, 3, 3>" % (32 + i, i % 2), "ms": 30.0 / (1 + i), "launches": 10 + i,
             "flops": 8.64e10 * (10 + i) / (1 + i), "bytes": 1.08e8 * (10 + i)} for i in range(60)]
    recs = bench.roofline_records(rows, bench.PEAK_FP32_MFMA_TFLOPS)
    assert len(recs) == 60
    short_sample = "CPU oracle joint step, B=16, 1 warm-up + 3 timed steps, median 65.6 s/step"
    long_sample = "oracle.nets_adv.joint_train_step (torch-CPU fp32 port of adversarial.py:839-882: 1 dis + clip + 1 gen), B=16 per domain, " \
                  "1 warm-up + 3 timed steps, median 65.61 s/step (all: 70.12, 65.61, 65.40, 66.02)"
    res = {"metric": "training slices/sec (256x256x3, B=16 per GPU) joint segmenter+GAN step (1 dis update on B MR + B CT, clip, 1 gen update on B CT)",
           "value": 166.66666666, "unit": "slices/s", "n_gpus": 8, "steps": 20, "warmup": 5, "ms_per_step": 95.98765432, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "BASELINE configs[3]: train_gan.py --phase train-gan joint step, B=16/GPU of each domain, f32, dropout .75, mask critic on",
                      "global_batch": 128, "per_gpu_batch": 16, "parallelism": "dp8", "final_loss": 0.123456789,
                      "comm": {"transport": "native-rccl", "rccl_version": 22105, "overlap": True, "buckets": 12,
                               "bucket_MB": [33.5] * 12, "distinct_bucket_sets": 2, "allreduce_MB_per_step": 315.0,
                               "dis_step": {"allreduce_MB": 150.0, "launch_order": list(range(12)), "exposed_ms": 0.41},
                               "gen_step": {"allreduce_MB": 165.0, "launch_order": list(range(12)), "exposed_ms": 0.39}}},
           "roofline": dict(recs[0]), "roofline_note": "x" * 700, "roofline_kernels": recs,
           "roofline_all_mfma_convs": {"achieved": 128.6, "peak": 157.3, "frac": 0.8175, "unit": "TFLOP/s", "ms_per_step": 81.2,
                                       "launches_per_step": 560.0, "probed_steps": 2},
           "step_algorithmic": {"tflop_per_step": 10.496, "achieved": 153.4123, "frac_of_mfma_peak": 0.97528},
           "segmenter_step": {"workload": "BASELINE configs[1]: ...", "value": 463.0, "unit": "slices/s", "ms_per_step": 34.5, "steps": 10,
                              "warmup": 2, "final_loss": 1.5},
           "bf16_step": {"workload": "BASELINE configs[4] arithmetic on the joint step: ...", "value": 463.0123, "unit": "slices/s", "ms_per_step": 34.5678,
                         "steps": 10, "warmup": 2, "final_loss": 0.51234},
           "cpu_baseline": {"value": 0.2441, "unit": "slices/s", "cores": 128, "cpu_model": "AMD EPYC 9575F 64-Core Processor", "kind": "port",
                            "batch": 16, "s_per_step": 65.61, "sample": short_sample, "sample_detail": long_sample, "value_B2": 0.1266, "sample_B2": long_sample},
           "kernels_file": "gpurun_out/bench_kernels_joint_f32_n8.json"}
    assert len(json.dumps(res)) > 20000                      # what round 3 printed
    rec = bench.compact_record(res)
    line = json.dumps(rec)
    assert len(line) < bench.MAX_LINE <= 1700 and "\n" not in line, len(line)
    back = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in back, k
    assert back["config"]["workload"].startswith("BASELINE configs[3]") and "roofline_kernels" not in back
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms"):
        assert k in back["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in back["cpu_baseline"], k
    assert "B=16" in back["cpu_baseline"]["sample"] and back["cpu_baseline"]["value_B2"] == 0.1266 and "sample_B2" not in back["cpu_baseline"]
    assert "sample_detail" not in back["cpu_baseline"]
    assert back["bf16_step"]["value"] == 463.0 and back["segmenter_step"]["value"] == 463.0 and back["step_algorithmic"]["achieved"] == 153.4
    assert abs(back["value"] - 166.7) < 0.05 and back["roofline"]["frac"] == float("%.4g" % recs[0]["frac"])
    assert back["config"]["comm"]["dis_step_exposed_ms"] == 0.41 and "launch_order" not in json.dumps(back)


def test_gpu_suite_wall_time_guard():
    """tests/conftest.py fails a -m gpu run that outgrows its wall-time budget (the driver stops it at 1200 s); CPU runs are exempt"""
    import conftest
    assert conftest.GPU_SUITE_BUDGET_S <= 1100                 # below the driver's 1200 s with room for pytest start-up and the import of torch
    assert conftest.suite_over_budget(901.0, 170, 900.0) and not conftest.suite_over_budget(899.0, 170, 900.0)
    assert not conftest.suite_over_budget(5000.0, 0, 900.0)          # the CPU suite ran no GPU test


def test_bench_prices_the_winograd_transforms_against_the_hbm_roof():
    """the transform kernels of the Winograd route carry no contraction (flops = 0 in pnp_prof_*): bench.roofline_records reports them as
    HBM-bound rows (algorithmic bytes / duration against 8 TB/s); a GEMM symbol next to them stays an MFMA row; a compact record whose
    dominant symbol is an HBM row still round-trips"""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    rows = [{"name": "wino_gemm_kernel<128, 128, 2, 2, 0>", "ms": 42.0, "launches": 200, "flops": 200 * 2.7e10, "bytes": 200 * 2.5e8},
            {"name": "wino_in_kernel", "ms": 6.0, "launches": 200, "flops": 0.0, "bytes": 200 * 1.45e8},
            {"name": "wino_out_kernel", "ms": 60.0, "launches": 200, "flops": 0.0, "bytes": 200 * 1.6e8}]
    recs = bench.roofline_records(rows, bench.PEAK_FP32_MFMA_TFLOPS)
    by = {r["kernel"]: r for r in recs}
    g, i = by["wino_gemm_kernel<128, 128, 2, 2, 0>"], by["wino_in_kernel"]
    assert g["bound"] == "mfma" and g["unit"] == "TFLOP/s" and abs(g["achieved"] - 200 * 2.7e10 / 42e-3 / 1e12) < 1e-6
    assert i["bound"] == "hbm" and i["unit"] == "GB/s" and i["peak"] == bench.PEAK_HBM_GBS
    assert abs(i["achieved"] - 200 * 1.45e8 / 6e-3 / 1e9) < 1e-3 and abs(i["frac"] - i["achieved"] / 8000.0) < 1e-9
    assert recs[0]["kernel"] == "wino_out_kernel" and recs[0]["bound"] == "hbm"          # sorted by time, whatever the bound
    # round 5: a Winograd GEMM row says that it counts EXECUTED flops and carries the same launch in the convolution's algorithmic flops
    # (x 2.25 for F(2x2): symbols <.., 0 / 1>; x 4 for F(4x4): <.., 2 / 3> and the 128 x 64 tile; filter gradient by its TILE parameter)
    assert g["flops_counted"] == "executed" and abs(g["achieved_algorithmic"] - 2.25 * g["achieved"]) < 1e-9 and i["flops_counted"] == "none"
    for name, f in (("wino_gemm_kernel<128, 128, 2, 2, 2>", 4.0), ("wino_gemm_kernel<128, 64, 2, 2, 3>", 4.0), ("wino_gemm_kernel<128, 128, 2, 2, 1>", 2.25),
                    ("wino_wgrad_gemm_kernel<128, 128, 2, 2, 4>", 4.0), ("wino_wgrad_gemm_kernel<128, 128, 2, 2, 2>", 2.25),
                    ("conv_taps_kernel<128, 128, 2, 2, 0, 3, 3>", 1.0)):
        assert bench.wino_alg_factor(name) == f, name
    d = bench.roofline_records([{"name": "conv_taps_kernel<128, 128, 2, 2, 0, 3, 3>", "ms": 1.0, "launches": 1, "flops": 1e11, "bytes": 1e8}], 157.3)[0]
    assert d["flops_counted"] == "algorithmic" and d["achieved_algorithmic"] == d["achieved"]
    assert bench.ALG_GFLOP_PER_SLICE["joint"] == 399.8 + 256.2 and bench.ALG_GFLOP_PER_SLICE["segmenter"] == 248.95
    rec = bench.compact_record({"metric": "m", "value": 1.0, "unit": "slices/s", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": 1.0,
                                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                                "config": {"workload": "w"}, "roofline": dict(recs[0])})
    back = json.loads(json.dumps(rec))
    assert back["roofline"]["bound"] == "hbm" and back["roofline"]["unit"] == "GB/s"

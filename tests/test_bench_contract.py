"""CPU: the bench line's contract, checked on the last line recorded under profiles/ (bench.py itself needs a GPU): the keys the driver
reads, and the internal consistency of the roofline / cpu_baseline objects (frac = achieved / peak, achieved = algorithmic bytes per launch
/ the dominant kernel's duration, throughput = terms per step / ms per step)."""
import glob
import json
import os
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_line():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default.json")))
    assert files, "no recorded bench line"
    return json.loads(open(files[-1]).read().strip().splitlines()[-1]), files[-1]


def test_recorded_bench_line_honours_the_contract():
    d, name = _last_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, (k, name)
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert "G1 MSM" in d["metric"] and "synthetic" in d["data"]
    # whole-job throughput: 2^20-term MSMs per second = 1000 / ms per step
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) / d["value"] < 0.01
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    # achieved = 128 B/term x 2^20 terms per k_accumulate launch / its duration
    assert abs(r["achieved"] - 128 * (1 << 20) / (r["avg_ms"] * 1e-3) / 1e9) / r["achieved"] < 0.01
    assert r["avg_ms"] < d["latency_ms_one_in_flight"] and 0 < r["frac"] < 1
    assert r["traffic"] is None or r["traffic"] > 128 * (1 << 20)          # counter traffic: the table method re-reads 13 rows per term
    v = d["valu_roofline"]
    assert abs(v["frac"] - v["achieved"] / v["peak"]) < 1e-3 and v["mixed_additions_per_launch"] == 13 * (1 << 20)
    c = d["cpu_baseline"]
    if c is not None:
        assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
        assert d["value"] / c["value"] > 20          # north_star: >= 20x the CPU path on the same box
    assert d["device_allocations_in_timed_region"] == 0


def test_traffic_figure_belongs_to_the_kernel_that_is_shipped():
    """`roofline.traffic` is read from profiles/traffic_accumulate.json, a counter measurement made at the commit the file names.  It describes HEAD's
    k_accumulate only if nothing that kernel is compiled from changed since: the commit must be an ancestor of HEAD and `git diff <commit> HEAD` must be
    empty for the kernel's sources (the kernel itself, the field and the group law it instantiates)."""
    import json
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("git") is None or not os.path.isdir(os.path.join(root, ".git")):
        pytest.skip("no git history here (the GPU box receives a snapshot without .git)")
    tr = json.load(open(os.path.join(root, "profiles", "traffic_accumulate.json")))
    commit = tr["commit"]
    git = lambda *a: subprocess.run(["git", "-C", root] + list(a), capture_output=True, text=True)
    assert git("cat-file", "-e", commit + "^{commit}").returncode == 0, "profiles/traffic_accumulate.json names commit %s, which this history does not contain" % commit
    assert git("merge-base", "--is-ancestor", commit, "HEAD").returncode == 0, "%s is not an ancestor of HEAD" % commit
    srcs = ["crypto_amd/csrc/msm_kernels.hip.h", "crypto_amd/csrc/fp30s.hip.h", "crypto_amd/csrc/ec29.hip.h", "crypto_amd/csrc/fp29.hip.h", "crypto_amd/csrc/dyn_chunk.hip.h"]
    d = git("diff", "--stat", commit, "HEAD", "--", *srcs)
    assert d.returncode == 0 and d.stdout.strip() == "", "k_accumulate's sources changed since the traffic measurement (%s): re-run tools/dev/round6_profiles.sh PART=msm\n%s" % (commit, d.stdout)

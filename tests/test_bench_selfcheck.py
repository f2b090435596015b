"""CPU tier: every committed bench line (profiles/rNN_bench_*.json, the driver's BENCH_rNN.json) must follow from its own
inputs - roofline.frac, hbm_traffic_frac and valu.frac are recomputed by tests/helpers.check_bench_line, the same function the GPU
tier applies to live output.  And tools/pmc_accumulate.py, which writes profiles/rNN_pmc_msm_accumulate.json, must reproduce the
summary from the raw per-pass tables."""
import glob
import json
import os
import subprocess
import sys

import pytest

from helpers import check_bench_line

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lines():
    out = []
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[3-9]_bench_*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r0[3-9]_kernel_trace_*_benchline.json"))):
        txt = open(p).read().strip()
        if txt.startswith("{"):
            out.append((os.path.basename(p), json.loads(txt)))
    for p in sorted(glob.glob(os.path.join(ROOT, "BENCH_r0[3-9].json"))):
        d = json.load(open(p))
        if isinstance(d.get("parsed"), dict):
            out.append((os.path.basename(p), d["parsed"]))
    return out


@pytest.mark.parametrize("name,line", _lines(), ids=[n for n, _ in _lines()])
def test_committed_bench_lines_follow_from_their_own_fields(name, line):
    if "roofline" not in line or not line["roofline"]:
        pytest.skip("no roofline object in this mode's line")
    check_bench_line(line)


def test_check_bench_line_catches_a_stale_fraction():
    name, line = next((n, l) for n, l in _lines() if l.get("roofline"))
    bad = json.loads(json.dumps(line))
    bad["roofline"]["frac"] *= 1.5
    with pytest.raises(AssertionError):
        check_bench_line(bad)
    bad = json.loads(json.dumps(line))
    bad["roofline"]["avg_launch_ms"] *= 0.5
    with pytest.raises(AssertionError):
        check_bench_line(bad)


def test_pmc_accumulate_summary_is_reproducible_from_the_raw_passes(tmp_path):
    """profiles/rNN_pmc_msm_accumulate.json is generator output: tools/pmc_accumulate.py over the per-pass JSON tables
    (tools/pmc_summary.py --json) of the same round gives the committed file again."""
    done = 0
    for summary in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[4-9]_pmc_msm_accumulate.json"))):
        tag = os.path.basename(summary)[:3]
        out = tmp_path / "again.json"
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_accumulate.py"), os.path.join(ROOT, "profiles"), tag, str(out)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert json.load(open(out)) == json.load(open(summary)), summary
        done += 1
    if not done:
        pytest.skip("no round >= 4 PMC summary committed yet")


def test_cpu_baseline_counts_the_cores_it_is_granted(tmp_path, monkeypatch):
    """bench_cpu.effective_cores: visible CPUs, affinity mask and the cgroup quota, whichever is smallest (the GPU boxes show 256
    CPUs behind `cpu.max` = 1600000 100000; rounds 1-4 reported the 256)."""
    import builtins
    import bench_cpu
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            p = tmp_path / "cpu.max"
            p.write_text("1600000 100000\n")
            return real_open(p, *a, **k)
        return real_open(path, *a, **k)

    monkeypatch.setattr(builtins, "open", fake_open)
    monkeypatch.setattr(bench_cpu.os, "cpu_count", lambda: 256)
    monkeypatch.setattr(bench_cpu.os, "sched_getaffinity", lambda pid: set(range(256)))
    n, info = bench_cpu.effective_cores()
    assert n == 16 and info["visible_cpus"] == 256 and info["cgroup_cpu_max"] == "1600000 100000"


def test_proof_algorithmic_bytes_follow_the_survey():
    """bench.proof_algorithmic_bytes = SURVEY.md section 8d's derivation: BN254 2^17 without BSB22 is 10 n 96 B of MSM (126 MB) +
    (12 n + 13 4n) 64 B of transforms (537 MB) + 14 4n 32 B of quotient pass (235 MB) + 50 MB = 0.95 GB; a BSB22 commitment adds one
    MSM, one pair of transforms and two vectors of the quotient pass; BLS12-381 pairs are 128 B."""
    import bench
    from algoplonk_amd import ecc
    a = bench.proof_algorithmic_bytes(ecc.BN254, 17, 0)
    assert (a["msm"], a["ntt"], a["quotient"], a["misc"]) == (125829120, 536870912, 234881024, 50000000) and a["total"] == 947581056
    b = bench.proof_algorithmic_bytes(ecc.BLS12_381, 21, 1)
    n = 1 << 21
    assert b["msm"] == 11 * n * 128 and b["ntt"] == (13 * n + 14 * 4 * n) * 64 and b["quotient"] == 16 * 4 * n * 32 and b["misc"] == 50000000 * 16
    assert bench.default_witnesses(bench.parse_args([])) == 16 and bench.default_witnesses(bench.parse_args(["--log-n", "21", "--inflight", "8"])) == 2
    assert bench.default_witnesses(bench.parse_args(["--inflight", "1"])) == 1

"""bench.py's host-side helpers (no GPU): the power probe's parsing of rocm-smi, the PMC-traffic lookup against the committed
profiles, and the refusal to run without a HIP device."""
import os
import stat
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SMI_TEXT = """

============================ ROCm System Management Interface ============================
================================= Current clock frequencies ==================================
GPU[0]		: mclk clock level: 0: (2000Mhz)
GPU[0]		: sclk clock level: 1: (1706Mhz)
=================================== Power Consumption ====================================
GPU[0]		: Current Socket Graphics Package Power (W): 1399.0
======================================= Power Cap ========================================
GPU[0]		: Max Graphics Package Power (W): 1400.0
================================== End of ROCm SMI Log ===================================
"""


def test_power_probe_parses_rocm_smi(tmp_path, monkeypatch):
    import bench
    fake = tmp_path / "rocm-smi"
    fake.write_text("#!/bin/sh\ncat <<'EOT'\n" + SMI_TEXT + "EOT\n")
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ.get("PATH", ""))
    steps, syncs = [0], [0]
    got = bench.power_probe(lambda: steps.__setitem__(0, steps[0] + 1), "a test load",
                            lambda: syncs.__setitem__(0, syncs[0] + 1), seconds=0.05)
    assert got["package_w"] == 1399.0 and got["cap_w"] == 1400.0 and got["sclk_mhz"] == 1706 and got["at_cap"] is True
    assert got["load"] == "a test load" and got["steps_under_load"] == steps[0] > 0 and syncs[0] > 0


def test_power_probe_reads_the_hwmon_files(tmp_path, monkeypatch):
    import bench
    for name, value in (("power1_input", 1398000000), ("freq1_input", 1534000000), ("power1_cap", 1400000000)):
        (tmp_path / name).write_text(f"{value}\n")
    monkeypatch.setattr(bench, "_hwmon_of", lambda dev_index: str(tmp_path))
    got = bench.power_probe(lambda: None, "a test load", lambda: None, seconds=0.02)
    assert got["package_w"] == 1398.0 and got["cap_w"] == 1400.0 and got["sclk_mhz"] == 1534 and got["at_cap"] is True
    assert got["source"].startswith("hwmon") and got["steps_under_load"] > 0


def test_power_probe_without_rocm_smi_output(tmp_path, monkeypatch):
    import bench
    fake = tmp_path / "rocm-smi"
    fake.write_text("#!/bin/sh\necho nothing useful\n")
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ.get("PATH", ""))
    assert bench.power_probe(lambda: None, "x", lambda: None, seconds=0.01) is None


def test_pmc_traffic_accepts_only_the_launched_instantiation():
    import bench
    n, d = 10_000_000, 512
    got = bench.pmc_traffic(n, d, 1, "ip_scan", "flat_scan_kernel<64, 2, 2, 0, 0, true, 0, false, false>")
    assert got["bytes"] is not None and abs(got["bytes"] / (n * d * 4) - 1.0) < 0.01, got
    assert got["source"]["file"].endswith("_pmc_summary.json") and "flat_scan_kernel" in got["source"]["kernel"]
    # the fp16-shadow pass streams 2 bytes per element, three main launches per corpus pass
    got = bench.pmc_traffic(n, d, 128, "ip_scan_half", "flat_scan_h16_kernel<32, 32, 4, 2>")
    assert got["bytes"] is not None and abs(3 * got["bytes"] / (n * d * 2) - 1.0) < 0.05, got
    # another instantiation of the same kernel: refused, and the refusal says why
    got = bench.pmc_traffic(n, d, 1, "ip_scan", "flat_scan_kernel<32, 2, 2, 0, 0, true, 0, false, false>")
    assert got["bytes"] is None and "refused" in got["source"]
    assert bench.pmc_traffic(n, d, 1, "ip_scan", "")["bytes"] is None


def test_bench_refuses_to_run_without_a_hip_device():
    import torch
    if torch.cuda.is_available():
        return
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "no CPU fallback" in (p.stderr + p.stdout)


def test_cpu_baseline_times_the_whole_corpus_unscaled():
    """cpu_baseline streams EVERY row of the resident corpus to the host (round-4 review: the first 1M of 10M rows were timed
    and the rate scaled x0.1) and times the oracle over all of it; the streamed, NUMA-placed copy holds the rows bit for bit."""
    import numpy as np
    import bench
    from oracle import flat

    n, d, k = 30_000, 64, 5
    x = flat.synth(n, d, 1234)
    flat.normalize_l2(x)
    q = flat.synth(4, d, 5678)
    flat.normalize_l2(q)

    class Idx:
        fetched = 0

        def get_rows(self, row0, m, out=None):
            self.fetched += m
            out[...] = x[row0:row0 + m]
            return out

    idx = Idx()
    got = bench.cpu_baseline(None, idx, d, k, q, n, budget_s=0.6)
    assert idx.fetched == n
    assert got["kind"] == "port" and got["cores"] == 1 and got["value"] > 0 and got["multithread_value"] > 0
    assert f"{n} of {n} rows" in got["sample"] and "nothing scaled" in got["sample"]
    dst = np.empty((n, d), dtype=np.float32)
    for b in range(0, n, 7000):
        flat.first_touch_copy_block(dst, np.ascontiguousarray(x[b:b + 7000]), 3, b)
    assert dst.tobytes() == x.tobytes()

"""-m gpu: BASELINE.json configs[2], [3], [4] at FULL size on the device, checked against the CPU restatements.

These run the very functions `python bench.py --config 3|4|5` runs (one timed step, no soak) and assert what those lines only
report: every reply verdict and exit count of cfg 3 against oracle/c/oracle.c and every read answer against
oracle.collective.max_timestamped_value; one 125,000-write call of cfg 4 (26.6 M signature packets, 256 replicas) against the C
restatement on its distinct writes and against the corpus' construction; 10,000 operations of every threshold scheme of cfg 5
against oracle/c/threshold.c byte for byte.  cfg 2 at full size: tests/test_gpu_parity.py
(test_cfg2_full_size_identity_against_the_c_oracle)."""
import pytest

import bench

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def D(gpu_ctx):          # gpu_ctx first: torch's HIP runtime is loaded before the library's (see conftest)
    d = bench.Dist(bench.parse_args(["--config", "2"]))
    yield d
    d.close()


def _args(cfg, *extra):
    return bench.parse_args(["--config", str(cfg), "--steps", "1", "--warmup", "0", "--soak-seconds", "0", "--cpu-budget", "8"] + list(extra))


def test_cfg3_full_size_reply_verdicts_and_read_answers(D):
    out = bench.bench_cfg3(_args(3), D)
    cfg = out["config"]
    assert cfg["replicas"] == 64 and cfg["variables_per_gpu"] == 10000 and 95000 < cfg["replies_per_gpu"] < 101000
    ops = out["pubkey_ops_per_step_per_gpu"]
    assert ops["rsa"] > 1500000 and ops["dsa"] > 1500000              # mixed: about half of ~4.3 M verifications each
    cpu = out["cpu_baseline"]
    assert cpu["gpu_verdicts_identical_to_cpu"] is True               # error byte and exit count of every reply (oracle/c/oracle.c)
    assert cpu["read_answers_identical_to_oracle"] is True            # <value, t> of every variable (oracle/collective.py)
    assert out["allgather"]["rows_consistent"] and 0.9 < out["reads_answered_fraction"] <= 1.0
    assert 0.9 < out["replies_accepted_fraction"] < 1.0               # the corpus' rejected replies are really rejected
    assert out["reference_pubkey_ops_per_step_per_gpu"] <= ops["rsa"] + ops["dsa"]


def test_cfg4_one_resident_call_of_125000_writes(D):
    out = bench.bench_cfg4(_args(4, "--items", "125000", "--chunk", "125000"), D)
    cfg = out["config"]
    assert cfg["replicas"] == 256 and cfg["writes_per_call"] == 125000 and cfg["tiles"] == 50 and cfg["sigs_per_call"] > 25000000
    assert out["verdicts_match_construction"] is True
    assert out["cpu_baseline"]["gpu_verdicts_identical_to_cpu"] is True     # every tile's verdicts and exit counts = the C restatement's
    assert out["allgather"]["rows_consistent"] and 0.95 < out["sufficient_fraction"] < 1.0
    assert out["reference_pubkey_ops_per_call"] <= out["pubkey_ops_per_call"] < cfg["sigs_per_call"]


def test_cfg5_full_size_every_scheme_against_the_c_restatement(D):
    out = bench.bench_cfg5(_args(5), D)
    assert out["config"]["ops_per_scheme"] == 10000 and out["config"]["schemes"] == 3
    assert out["cpu_baseline"]["gpu_results_identical_to_cpu"] is True
    assert set(out["kernel_ms"]) == {"rsa_combine_n10", "sss_calculate_secret_k7_2048", "dsa_calculate_s_2t8_q256", "dsa_calculate_r_2t8_2048_256"}

"""CPU-side checks added in round 6: the bench headline stays a line the driver can parse (BENCH_r05.parsed was
null because the default line had grown to 25 KB), and the defaults stay the short run."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

HEADLINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline", "cpu_baseline", "conformal", "gpu_sensors", "extra_file")


def _fake_inputs(n_extra):
    import bench_extras as bx
    stage = lambda b, u: dict(bound=b, achieved=5933.7, unit=u, frac=0.7417, launches=25, ms_per_step=10.903)   # noqa: E731
    roof = dict(bound="mfma", kernel="conv_wg3s_kernel<64>", achieved=108.43, peak=157.3, unit="TFLOP/s", frac=0.6893,
                traffic=12552167424, traffic_source="r6_pmc_traffic.json", algorithmic_bytes_per_launch=4295409664,
                effective_tflops=365.81, mfma_share_of_direct_form=0.2963, launches_per_step=10, avg_launch_ms=6.0861, share_of_step=0.261,
                stages={k: stage("mfma" if i % 2 else "hbm", "TFLOP/s" if i % 2 else "GB/s")
                        for i, k in enumerate(("ta_block_w64", "gn_apply_silu", "la_block", "tattn_core", "temporal_attention_all",
                                               "conv_gn_silu_block", "ta_block_w128"))},
                stage_peaks={"TFLOP/s": 157.3, "GB/s": 8000.0}, stage_sum_ms=233.123,
                top_kernels=[[f"conv_rh_kernel<64,128,2,2,7,true>_{i}", 12, 43.912, 115.44] for i in range(6)])
    cpu = dict(value=3.3e-4, unit="trajectories/s", cores=16, kind="port", cpu_model="AMD EPYC 9575F 64-Core Processor",
               logical_cpus_visible=256, sample="2 guided p_sample steps (after 1 warm-up) at B=1 of the same workload, 3029 ms/step, "
                                                "extrapolated x1000 steps per trajectory", all_physical_cores={"x": 1})
    clocks = dict(source="hwmon", samples=100, sclk_mhz_min=2100, sclk_mhz_median=2280, sclk_mhz_max=2400, sclk_mhz_mean=2275,
                  power_w_mean=1238, power_w_max=1400)
    cf = dict(Q=0.0123456789, n_cal_per_rank=200, n_cal=200, alpha=0.04, score_kernel_ms=0.496, allgather_quantile_ms=0.577,
              source="x" * 300)
    extra = {k: dict(workload="w" * 120, why=v[3], ms_per_step=11.2, value=22.8, roofline=dict(frac=0.58, kernel="k" * 40),
                     stages={"s": 1}, all_kernels={"k" * 40: {"ms": 1.0} for _ in range(1)})
             for k, v in list(bx.EXTRA_WORKLOADS.items())[:n_extra]}
    extra.update(calibration=dict(what="c" * 300, ms_per_step=95.1), strawman=dict(what="s" * 200, ms_per_step=4000.0, hip_speedup_per_trajectory=135.0),
                 finetune_step=dict(what="f" * 200, hip_ms=91.8, hip_graph_ms=88.0), phases_s={"a": 1.0})
    W = dict(desc="C4: 2D smoke Unet3D_with_Conv3D dim=64 (1,2,4) state (B,32,7,64,64), B=64 per GPU, guided 1000-step DDPM, conformal quantile on")
    a = argparse.Namespace(steps=20, warmup=5, precision="fp32")
    return a, W, roof, cpu, clocks, cf, extra


def test_headline_is_compact_and_complete():
    """the ONE stdout line: <= 6144 bytes with every extra workload riding along, json round trip, contract keys present"""
    import bench
    a, W, roof, cpu, clocks, cf, extra = _fake_inputs(8)
    out = bench.headline(a, 1, 64, 0.2742, 233.4, W, roof, cpu, clocks, cf, None, 1, extra, "bench_extra.json")
    line = json.dumps(out)
    assert len(line) <= bench.HEADLINE_MAX_BYTES == 6144, len(line)
    back = json.loads(line)
    for k in HEADLINE_KEYS:
        assert k in back, k
    assert back["metric"].startswith("sampled control trajectories/sec")
    assert back["config"]["workload"].startswith("C4") and "model" not in back["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "stages"):
        assert k in back["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in back["cpu_baseline"], k
    assert "all_physical_cores" not in back["cpu_baseline"]
    # the extras ride as one small record each; their full blocks are in the side file only
    assert set(back["extra"]["c2"]) == {"ms_per_step", "value", "frac"}
    assert "what" not in back["extra"]["calibration"] and "phases_s" not in back["extra"]


def test_headline_sheds_optional_keys_rather_than_overflow():
    import bench
    a, W, roof, cpu, clocks, cf, extra = _fake_inputs(8)
    roof["stages"] = {f"stage_{i}": dict(bound="mfma", achieved=1.0, unit="TFLOP/s", frac=0.5, launches=1, ms_per_step=1.0, pad="p" * 60)
                      for i in range(32)}
    out = bench.headline(a, 1, 64, 0.27, 233.4, W, roof, cpu, clocks, cf, None, 1, extra, "bench_extra.json")
    assert len(json.dumps(out)) <= bench.HEADLINE_MAX_BYTES
    assert "roofline" in out and "cpu_baseline" in out and "extra" not in out


def test_bench_defaults_are_the_short_run():
    """default extras = the two other single-GPU BASELINE configs only; the 128-core CPU leg and the shipped-width / shard8 /
    fine-tuning extras are opt-in (285 s of driver time, 104 s of it the all-cores CPU leg)"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"--extra-workloads", default="c2,c3"' in src
    assert '"--cpu-all-cores", action="store_true"' in src and '"--all-extras", action="store_true"' in src
    import bench_extras as bx
    assert set(bx.ALL_EXTRA_WORKLOADS.split(",")) <= set(bx.EXTRA_WORKLOADS)

"""world_size-2 coverage of the N>1 path on CPU (gloo): rank sharding as the reference script does it,
end-of-job image collation and the max-over-ranks timing reduction."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffusion_uncertainty_amd.distributed import gather_images, max_over_ranks, shard_range
    from diffusion_uncertainty_amd.utils import starting_points
    X, y = starting_points(7, (3, 4, 4))                     # 7 samples over 2 ranks: n = 3, remainder dropped
    lo, hi = shard_range(7, rank, world)
    imgs = (X[lo:hi].clamp(-1, 1) * 127 + 128).to(torch.uint8)
    allimgs = gather_images(imgs)
    t = max_over_ranks(1.0 + rank, "cpu")
    dist.barrier()
    q.put((rank, lo, hi, allimgs.numpy(), t))        # numpy: pickled by value (a torch tensor travels as an fd the
                                                     # worker may close on exit before the parent receives it)
    dist.destroy_process_group()


def test_two_rank_sharding_and_collation():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from diffusion_uncertainty_amd.utils import starting_points
    X, _ = starting_points(7, (3, 4, 4))
    ref = (X[:6].clamp(-1, 1) * 127 + 128).to(torch.uint8).reshape(2, 3, 3, 4, 4)
    assert [(r[1], r[2]) for r in res] == [(0, 3), (3, 6)]
    for r in res:
        assert torch.equal(torch.from_numpy(r[3]), ref)                        # every rank holds the full collation, in rank order
        assert r[4] == 2.0                                   # max over ranks


def test_single_process_fallbacks():
    from diffusion_uncertainty_amd.distributed import gather_images, max_over_ranks, shard_range
    x = torch.zeros(2, 3, 4, 4, dtype=torch.uint8)
    assert gather_images(x).shape == (1, 2, 3, 4, 4)
    assert max_over_ranks(1.5, "cpu") == 1.5
    assert shard_range(1024, 3, 8) == (384, 512)


def test_bench_launcher_starts_ranks_itself():
    """`python bench.py --gpus 2` with no rendezvous in the environment must start two ranks itself (--gpus
    was a no-op) and rank 0 must print ONE JSON line with n_gpus = 2.  CPU rehearsal: gloo backend, stub job."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "0",
                          "--backend", "gloo", "--stub-job"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["ranks_seen"] == 2 and r["steps"] == 20 and r["backend"] == "gloo"


def test_bench_launcher_world8_rehearsal_shards_1024_images():
    """BASELINE configs[3] is 1 024 images over 8 ranks.  The same launch path (`python bench.py --gpus 8`: eight
    fresh ranks over a 127.0.0.1 rendezvous, rank-sliced inputs, barrier bracket, one all_gather per job, max over ranks) rehearsed on the
    CPU with gloo and the stub job: every rank takes 128 images, the collation holds all 8 rank blocks in rank order."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "10", "--warmup", "0", "--backend", "gloo",
                          "--stub-job", "--stub-images", "1024"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["ranks_seen"] == 8 and r["images"] == 1024 and r["images_per_rank"] == 128 and r["collated_ok"] is True


def test_bench_short_line_is_short_and_complete():
    """the LAST line bench.py prints is the contract record in short form (< 2 KB: never truncated by a tail reader);
    the long legs go to a side file and a line of their own."""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    long_text = "x" * 3000
    line = {"metric": "images/sec @ ADM ImageNet-128, 50 steps, M=5, bs=128; 1/2/4/8 GPU", "value": 9.0, "unit": "images/sec", "n_gpus": 1, "steps": 20,
            "warmup": 5, "ms_per_step": 1400.0, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16x3 (" + long_text + ")",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: ADM ImageNet-128, 50 DDIM steps, M=5, uncertainty_zigzag_centered, num_zigzag=3, UC window (start 40, 10 steps), batch 128 per GPU",
                       "global_batch": 128, "parallelism": "independent batches x1 (" + long_text + ")", "ranks_seen": 1, "precision": "f16x3",
                       "executed_forwards_per_image": 100.0, "reference_forwards_per_image": 200, "zigzag_mode": "skip_dead_zigzag (" + long_text + ")",
                       "parity": long_text},
            "end_to_end_tflops": 550.0, "value_at_reference_forward_count": 4.5,
            "roofline": {"bound": "mfma", "achieved": 820.0, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.328, "traffic": 2.9e9, "traffic_source": "profiles/x.json",
                         "kernel": "k_wconv4 / k_wconv (csrc/...): " + long_text, "launches": 14200, "flops_per_launch": 1.0e12, "ms_per_launch": 1.22,
                         "executed_frac": 0.51, "executed_products_per_algorithmic_product": 1.55, "note": long_text},
            "cpu_baseline": {"value": 0.0143, "value_at_median": 0.013, "unit": "images/sec", "cores": 16, "kind": "port", "sample": "oracle ADM-128: " + long_text,
                             "gpu_over_cpu_same_forward_count": 310.0},
            "checked": {"ok": True, "max_abs_diff": 0.0, "bit_identical": 3, "of": 3, "what": long_text, "conv3x3_forms": {"a": long_text}},
            "other_precision_modes": [{"precision": "fp32", "value": 2.9, "dtype": long_text}], "other_configs": [{"config": "U-ViT " + long_text, "value": 12.2}]}
    sh = b.short_line(line, os.path.join(root, "gpurun_out", "bench_legs.json"))
    text = json.dumps(sh)
    assert len(text) < 2048, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "checked", "full_record"):
        assert k in sh, k
    assert sh["roofline"]["frac"] == 0.328 and sh["cpu_baseline"]["cores"] == 16 and sh["config"]["workload"].startswith("BASELINE configs[1]")
    assert sh["roofline"]["kernel"] == "k_wconv4 / k_wconv (csrc/...)" and sh["dtype"] == "f16x3"


def test_bench_launcher_propagates_failure():
    """A failing rank must make the launcher exit non-zero (no silent single-rank fallback)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    # without --stub-job the ranks need a GPU: on this CPU box every rank raises SystemExit("bench.py needs a GPU ...")
    if torch.cuda.is_available():
        return
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10"], capture_output=True,
                         text=True, timeout=300, env=env)
    assert out.returncode != 0


def test_bench_steps_round_up_to_whole_jobs():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    a = b.parse_args(["--steps", "20", "--warmup", "5"])
    assert a.precision == "f16x3" and a.M == 5 and a.steps == 20        # driver invocation: 2 whole jobs of the fp32-equivalent mode
    assert b.parse_args(["--config", "3"]).M == 10                    # BASELINE configs[3]: M = 10, 128 images per GPU
    assert b.MODES["f16x3"][1] == 3.0 and b.STEP_TIMESTEPS * 10 == 50

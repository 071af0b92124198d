"""BASELINE config 5 shape: full rjMCMC on S synthetic FDEM soundings x n iterations with birth / death / perturb moves,
one process per GPU (launch N > 1 ranks with torch.distributed.run), soundings sharded by the reference's block rule,
posterior summaries and layer-count histograms gathered on rank 0 (RCCL all-gather; GBP_BENCH_BACKEND=gloo for the
single-GPU functional check of the N > 1 path).  The random streams are keyed by the global sounding index, so the
gathered result does not depend on the number of GPUs.

    python scripts/run_config5.py [--soundings 8192] [--iterations 10000] [--layers 4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 \\
        scripts/run_config5.py --soundings 8192 --iterations 10000
"""
import argparse, json, os, sys, time, hashlib
import numpy as np
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import warnings; warnings.filterwarnings("ignore")

ap = argparse.ArgumentParser()
ap.add_argument("--soundings", type=int, default=8192)
ap.add_argument("--iterations", type=int, default=10000)
ap.add_argument("--layers", type=int, default=4, help="layers of the synthetic true models")
ap.add_argument("--seed", type=int, default=2026)
ap.add_argument("--forward-waves", type=int, default=2, help="pinned waves per workgroup of the forward kernels (0 = adaptive)")
ap.add_argument("--hankel-eps-ppm", type=float, default=None, help="abscissa window budget (default 1e-10 ppm; 0 = all 120 / 140 abscissae)")
ap.add_argument("--reference-jacobian", action="store_true", help="use the reference's Jacobian expression in the proposals")
args = ap.parse_args()
rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
dev_index = local % torch.cuda.device_count()
torch.cuda.set_device(dev_index)
device = torch.device("cuda", dev_index)
if world > 1:
    backend = os.environ.get("GBP_BENCH_BACKEND", "nccl")
    dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": device} if backend == "nccl" else {}))

from geobipy_amd import DeviceChains, FdemBatch, synthetic, _lib
from geobipy_amd.distributed import SummaryGather, shard

S, n_it = args.soundings, args.iterations
system = synthetic.syn10_system()
start, Bloc = shard(S, rank, world)
# the whole survey is drawn with one seed and sliced, so the workload does not depend on the number of ranks
nl, sigma_true, thk, height = synthetic.draw_models(S, args.layers, seed=synthetic.SEED + 5)
sl = slice(start, start + Bloc)
clean = FdemBatch(system, nl[sl], sigma_true[sl], thk[sl], height[sl], device=device,
                  waves=args.forward_waves).forward().cpu().numpy()      # the synthetic data must not depend on the shard size either
noise = np.random.Generator(np.random.PCG64DXSM(synthetic.SEED + 6)).normal(size=(S, clean.shape[1]))[sl]
data = clean + noise * np.sqrt((0.05 * clean) ** 2 + 5.0 ** 2)
options = dict(solve_gradient=True, maximum_number_of_layers=30, minimum_depth=1.0, maximum_depth=150.0, minimum_thickness=1.0,
               initial_relative_error=0.05, minimum_relative_error=0.001, maximum_relative_error=0.5, initial_additive_error=5.0,
               minimum_additive_error=3.0, maximum_additive_error=20.0, relative_error_proposal_variance=1e-6,
               additive_error_proposal_variance=1e-6, probability_of_birth=1.0 / 6.0, probability_of_death=1.0 / 6.0,
               probability_of_perturb=1.0 / 6.0, probability_of_no_change=0.5)
dc = DeviceChains(system, height[sl], data, seed=args.seed, exact_jacobian=not args.reference_jacobian, first_chain=start,
                  forward_waves=args.forward_waves, device=device, hankel_eps_ppm=args.hankel_eps_ppm,
                  **options)
m0 = dc.misfit.clone()
K = dc.K
gather = SummaryGather(S, 6 + K + 1, device)


def sync():
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()


sync(); t0 = time.perf_counter()
dc.run(n_it // 2, accumulate=False)             # burn-in
dc.run(n_it - n_it // 2, accumulate=True)
sm = dc.summaries()
cols = [sm[:, i] for i in range(6)] + [dc.k_hist[:, i].to(torch.float64) for i in range(K + 1)]
out = gather.finish(gather.launch(*cols))       # the one exchange of the job: [S, 6 + K + 1] on rank 0
sync(); dt = time.perf_counter() - t0
t = torch.tensor([dt], dtype=torch.float64, device=device)
if world > 1:
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
if rank == 0:
    o = out.cpu().numpy()
    print(json.dumps({
        "workload": f"rjMCMC {S} soundings x {n_it} iterations, syn10 system (10 frequencies), {args.layers}-layer true models, K<=30",
        "n_gpus": world, "seconds": round(t.item(), 3), "chain_iterations_per_s": round(S * n_it / t.item()),
        "acceptance": round(float(o[:, 4].mean()), 3), "mean_layers": round(float(o[:, 3].mean()), 3),
        "median_misfit_20ch": round(float(np.median(o[:, 0])), 2), "median_start_misfit": round(float(m0.median()), 1),
        "layer_count_posterior": [round(float(v), 4) for v in (o[:, 6:].sum(axis=0) / o[:, 6:].sum())[:10]],
        "gathered_sha1": hashlib.sha1(np.ascontiguousarray(o).tobytes()).hexdigest()[:16],
        "jacobian": "reference expression" if args.reference_jacobian else "exact"}))
if world > 1:
    dist.destroy_process_group()

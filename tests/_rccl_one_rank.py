"""Worker of tests/test_gpu_rccl.py: ONE rank, backend "nccl" (= RCCL on ROCm), on the one GPU of the test box.

Started as a fresh child process (the process group is initialised before anything else touches the GPU).  Drives the
partitioned-cloud step of wlsqm.sharded.HaloCloudSolver with a loop-back halo, so that exchange_begin / exchange_end run their
RCCL branch — device tensors, all_to_all_single with uneven-split lists on the side stream, halo slots filled under the interior
fits — exactly as on an N-GPU node, and checks an RCCL all-reduce.  Writes <out>.json."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))


def main():
    out, port = sys.argv[1], int(sys.argv[2])
    import numpy as np
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    import synth
    import wlsqm.hip as whip
    from wlsqm.sharded import HaloCloudSolver
    res = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    x = torch.full((1 << 20,), 3.0, dtype=torch.float64, device=dev)
    dist.all_reduce(x)
    res["allreduce_sum_ok"] = bool((x == 3.0).all().item())

    N, k = 200_000, 40
    S = synth.halton(N, 3)
    S = torch.from_numpy(np.ascontiguousarray(S[synth.morton_order(S)])).to(dev)
    F = torch.sin(np.pi * S[:, 0]) * torch.cos(np.pi * S[:, 1]) * torch.exp(S[:, 2])

    def run(loopback, steps=4):
        s = HaloCloudSolver(3, S, k, order=2, knowns=0, weighting_method=2, device=dev)
        s.set_own_values_from_global(F)
        idx = torch.arange(0, s.n_own, 5, device=dev)
        if loopback:
            s.install_loopback_halo(idx)
            assert not s._host_stage and s._comm_stream is not None
        seen = []
        for _ in range(steps):
            fi = s.step()
            if loopback:
                torch.cuda.synchronize()
                seen.append(bool(torch.equal(s.values[s.n_own:], s.values[idx])))      # halo slots = what the owner held at exchange time
            s.values[: s.n_own] = fi[:, 0] + 1e-7 * (fi[:, 4] + fi[:, 6] + fi[:, 8])
        torch.cuda.synchronize()
        return s.values[: s.n_own].cpu().numpy(), seen, int(idx.numel())
    plain, _, _ = run(False)
    looped, seen, m = run(True)
    res.update({"kernel": whip.last_kernel(), "halo_values_per_step": m, "halo_slots_match_every_step": all(seen) and len(seen) == 4,
                "owned_values_bit_identical_to_the_run_without_exchange": bool(np.array_equal(plain, looped))})
    dist.barrier()
    dist.destroy_process_group()
    json.dump(res, open(out, "w"))


if __name__ == "__main__":
    main()

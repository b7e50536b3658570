"""Can the per-layer end-point exchange (dist.EndpointExchange: broadcasts of projected K / V^T on a side stream) be CAPTURED into
the pass's hipGraph on this ROCm / RCCL?  (VERDICT r5 next #9.)  Run under gpurun: one rank over nccl (= RCCL) on cuda:0 — with one
rank RCCL still goes through its enqueue path, which is what a capture has to record — or under torch.distributed.run with N ranks.

Prints one line per probe: `capture <what>: ok | FAILED <error>` and whether a replay reproduces the eager result."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    x = torch.arange(1 << 20, device=dev, dtype=torch.float32) * (rank + 1)
    outs = [torch.empty_like(x) for _ in range(world)]
    dist.broadcast(x, src=0)                      # eager warm-up: communicator creation is not capturable
    dist.all_gather(outs, x)
    torch.cuda.synchronize()

    def probe(name, fn, check):
        side = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        try:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.graph(g, stream=side):
                fn()
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            print(f"[rank {rank}] capture {name}: ok, replay reproduces the eager result: {bool(check())}", flush=True)
        except Exception as e:                    # noqa: BLE001
            print(f"[rank {rank}] capture {name}: FAILED {type(e).__name__}: {str(e)[:300]}", flush=True)
            try:
                torch.cuda.synchronize()
            except Exception as e2:               # noqa: BLE001
                print(f"[rank {rank}]   device state after the failure: {str(e2)[:200]}", flush=True)

    y = torch.full_like(x, -1.0) if rank else x.clone() * 3
    ref = x.clone() * 3
    if world > 1:
        dist.broadcast(ref, src=0)
    probe("broadcast", lambda: dist.broadcast(y, src=0), lambda: torch.equal(y, ref))
    probe("all_gather", lambda: dist.all_gather(outs, x), lambda: all(torch.isfinite(o).all() for o in outs))

    # the exchange itself: dist.EndpointExchange.exchange_async on projected keys / values, inside a captured pass
    from aid_amd import dist as adist
    n = 4
    sh = adist.owned_shard(n * world, world, rank)
    ex = adist.EndpointExchange(n * world, world, rank)
    k = torch.randn(sh.n_local + 2, 64, 128, device=dev, dtype=torch.bfloat16)
    vt = torch.randn(sh.n_local + 2, 128, 64, device=dev, dtype=torch.bfloat16)
    pend = ex.exchange_async(k, vt, sh.n_local)
    pend.wait()
    torch.cuda.synchronize()
    want = (k.clone(), vt.clone())

    def exchange():
        p = ex.exchange_async(k, vt, sh.n_local)
        p.wait()
    probe("EndpointExchange.exchange_async (+ wait)", exchange, lambda: torch.equal(k, want[0]) and torch.equal(vt, want[1]))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Can two RCCL ranks share the one GPU of the test box?  (If so the slab engine's real
transport can be rehearsed there; if not, RCCL says why.)  Starts two children on cuda:0,
each does one isend/irecv pair over backend "nccl" and prints what happened."""
import datetime
import os
import subprocess
import sys


def child(rank, world):
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev,
                                timeout=datetime.timedelta(seconds=60))
        a = torch.full((1024,), float(rank + 1), device=dev)
        b = torch.zeros(1024, device=dev)
        peer = 1 - rank
        ops = [dist.P2POp(dist.isend, a, peer), dist.P2POp(dist.irecv, b, peer)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        torch.cuda.synchronize()
        print("rank", rank, "received", float(b[0]), flush=True)
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        print("rank", rank, "failed:", str(e)[:600].replace("\n", " | "), flush=True)
        sys.exit(3)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1]), 2)
    else:
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update({k: v for k, v in (a.split("=", 1) for a in os.environ.get("PROBE_ENV", "").split() if "=" in a)})
        ps = [subprocess.Popen([sys.executable, __file__, str(r)], env=env) for r in range(2)]
        codes = []
        for p in ps:
            try:
                codes.append(p.wait(timeout=150))
            except subprocess.TimeoutExpired:
                p.kill()
                codes.append("timeout")
        print("exit codes", codes)

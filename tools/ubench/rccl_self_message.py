#!/usr/bin/env python3
"""Does a large message of one RCCL rank to ITSELF come back intact?  (llcomp_amd/sharding.py sends messages beyond 1 GiB in
rounds because a 2 GB self-message came back damaged in round 3; this is that exchange without any code of this repository:
torch.distributed, backend nccl (= RCCL), world size 1, all_to_all_single on uint8 device tensors.)

    python tools/ubench/rccl_self_message.py [sizes in MiB ...]        (one GPU; prints one verdict per size)
"""
import os
import socket
import sys

import torch
import torch.distributed as dist


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [512, 1024, 1536, 2047, 2048, 2049, 3072]
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
    sk.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    print("torch", torch.__version__, "RCCL", ".".join(str(v) for v in torch.cuda.nccl.version()), flush=True)
    bad = 0
    for mib in sizes:
        n = mib << 20
        g = torch.Generator(device="cuda").manual_seed(mib)
        src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda", generator=g)
        for what in ("all_to_all_single", "send_recv_self"):
            dst = torch.zeros(n, dtype=torch.uint8, device="cuda")
            try:
                if what == "all_to_all_single":
                    dist.all_to_all_single(dst, src, output_split_sizes=[n], input_split_sizes=[n])
                else:
                    ops = [dist.P2POp(dist.isend, src, 0), dist.P2POp(dist.irecv, dst, 0)]
                    for r in dist.batch_isend_irecv(ops):
                        r.wait()
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001
                print(f"{mib:5d} MiB {what:18s}: raised {type(e).__name__}: {str(e)[:160]}", flush=True)
                bad += 1
                continue
            n_bad, first, last, zeros = 0, -1, -1, 0
            step = 256 << 20  # compared piecewise: torch.nonzero on 2^31 elements overflows its own index arithmetic
            for lo in range(0, n, step):
                neq = dst[lo:lo + step] != src[lo:lo + step]
                k = int(neq.sum().item())
                if k:
                    idx = torch.nonzero(neq)[:, 0]
                    first = lo + int(idx[0].item()) if first < 0 else first
                    last = lo + int(idx[-1].item())
                    zeros += int((dst[lo:lo + step][idx] == 0).sum().item())
                    n_bad += k
                    del idx
                del neq
            if n_bad:
                print(f"{mib:5d} MiB {what:18s}: DAMAGED {n_bad} bytes differ, first at {first} (0x{first:x}), last at {last} (0x{last:x}), {zeros} of them read 0", flush=True)
                bad += 1
            else:
                print(f"{mib:5d} MiB {what:18s}: intact", flush=True)
            del dst
        del src
        torch.cuda.empty_cache()
    dist.destroy_process_group()
    print("verdict:", "damage seen" if bad else "every message intact")


if __name__ == "__main__":
    main()

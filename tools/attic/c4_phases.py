import sys, time, os
sys.path.insert(0, '.')
import torch, torch.distributed as dist
import llcomp_amd as mi
from llcomp_amd import sharding
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
dev=torch.device("cuda",0)
for images in (4, 8, 12, 16):
    sc = sharding.ShardedCodec(8192, 8192, 3, 512, 1, True, images=images, device=dev)
    band = torch.randint(0,256,(images,8192,8192,3),dtype=torch.uint8,device=dev)
    def T():
        torch.cuda.synchronize(); return time.perf_counter()
    sc.decode(sc.encode(band))
    t0=T(); payload, lens, total, status = sc.band.encode(band); t1=T()
    conts = sc.encode(band); t2=T()
    out = sc.decode(conts); t3=T()
    st = sc.band.decode(payload, int(total.item()), lens, out); t4=T()
    print(f"images {images}: local encode {1e3*(t1-t0):.1f} ms, full encode {1e3*(t2-t1):.1f}, full decode {1e3*(t3-t2):.1f}, local decode {1e3*(t4-t3):.1f}; workspace {sc.band.codec.workspace_bytes/1e9:.1f} GB", flush=True)
    del sc, band, conts, out, payload
    torch.cuda.empty_cache()

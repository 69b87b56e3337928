"""Multi-GPU sharding of ONE image over the ranks of a torch.distributed job (one process per GPU, RCCL over xGMI when
the backend is "nccl", gloo on CPU for tests).

Slices have fresh state and slice-local borders, so a horizontal band of whole tile rows, encoded as an image of its
own with the same tiling, yields exactly the slices of the full image.  Encode therefore needs no halo exchange and
exactly one exchange step: the variable-length gather of the per-rank containers to rank 0, where the host
concatenator (llcomp_mi_merge_bands) stitches them.  Decode is the mirror image: split_band + scatter, decode, gather
of raw rows.  The reference has no counterpart (single process, single stream): SURVEY.md 8(e).
"""
import numpy as np
import torch
import torch.distributed as dist

from . import FORMAT_SLICED, compress_image, decompress_image, merge_bands, probe, split_band


def band_rows(height, tile_h, world):
    """[(y0, y1)] per rank: contiguous bands of whole tile rows, as even as possible; ranks beyond the number of
    tile rows get an empty band (y0 == y1)."""
    tile_h = height if tile_h <= 0 or tile_h > height else tile_h
    nty = (height + tile_h - 1) // tile_h
    out, t = [], 0
    for r in range(world):
        cnt = nty // world + (1 if r < nty % world else 0)
        y0, y1 = min(height, t * tile_h), min(height, (t + cnt) * tile_h)
        out.append((y0, y1))
        t += cnt
    return out


def band_tile_rows(height, tile_h, world):
    tile_h = height if tile_h <= 0 or tile_h > height else tile_h
    return [((y0 + tile_h - 1) // tile_h, (y1 + tile_h - 1) // tile_h) for y0, y1 in band_rows(height, tile_h, world)]


def _device_for_backend():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def gather_bytes(payload: bytes, dst=0):
    """Variable-length gather of one byte string per rank to `dst` (None elsewhere): all_gather of the lengths, then
    point-to-point sends of the bodies (isend/irecv batch = ncclSend/ncclRecv group over xGMI on the nccl backend)."""
    rank, world, dev = dist.get_rank(), dist.get_world_size(), _device_for_backend()
    n = torch.tensor([len(payload)], dtype=torch.int64, device=dev)
    lens = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(lens, n)
    lens = [int(x.item()) for x in lens]
    if rank == dst:
        bufs = [None] * world
        reqs = []
        for r in range(world):
            if r == dst:
                bufs[r] = payload
            elif lens[r]:
                bufs[r] = torch.empty(lens[r], dtype=torch.uint8, device=dev)
                reqs.append(dist.irecv(bufs[r], src=r))
            else:
                bufs[r] = b""
        for q in reqs:
            q.wait()
        return [b if isinstance(b, (bytes, bytearray)) else b.cpu().numpy().tobytes() for b in bufs]
    if len(payload):
        t = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(dev)
        dist.isend(t, dst=dst).wait()
    return None


def scatter_bytes(parts, src=0):
    """Inverse of gather_bytes: rank `src` holds one byte string per rank, every rank returns its own."""
    rank, world, dev = dist.get_rank(), dist.get_world_size(), _device_for_backend()
    lens = torch.zeros(world, dtype=torch.int64, device=dev)
    if rank == src:
        lens = torch.tensor([len(p) for p in parts], dtype=torch.int64, device=dev)
    dist.broadcast(lens, src=src)
    mine = int(lens[rank].item())
    if rank == src:
        reqs = []
        keep = []
        for r in range(world):
            if r != src and len(parts[r]):
                t = torch.frombuffer(bytearray(parts[r]), dtype=torch.uint8).to(dev)
                keep.append(t)
                reqs.append(dist.isend(t, dst=r))
        for q in reqs:
            q.wait()
        return bytes(parts[src])
    if mine == 0:
        return b""
    buf = torch.empty(mine, dtype=torch.uint8, device=dev)
    dist.irecv(buf, src=src).wait()
    return buf.cpu().numpy().tobytes()


def encode_image_sharded(band, width, band_height, channels, *, tile_w=0, tile_h=0, planar=True, full_height=None,
                         encode_fn=None):
    """Every rank passes ITS band of the image (rows band_rows(...)[rank]); rank 0 gets the container of the whole
    image, the others None.  encode_fn(band, w, h, c, tile_w, tile_h, planar) -> container bytes defaults to the HIP
    path; tests inject another encoder to exercise the distributed plumbing without a GPU."""
    if encode_fn is None:
        def encode_fn(b, w, h, c, tw, th, pl):
            return compress_image(b, w, h, c, format=FORMAT_SLICED, tile_w=tw, tile_h=th, planar=pl)
    mine = encode_fn(band, width, band_height, channels, tile_w, tile_h, planar) if band_height > 0 else b""
    parts = gather_bytes(mine, dst=0)
    if dist.get_rank() != 0:
        return None
    return merge_bands([p for p in parts if len(p)])


def decode_image_sharded(container, *, decode_fn=None):
    """Rank 0 passes the container (others None); every rank decodes its band of tile rows; rank 0 returns the
    pixels (np.uint8 [h,w,c]), the others None."""
    rank, world = dist.get_rank(), dist.get_world_size()
    if decode_fn is None:
        def decode_fn(data):
            return decompress_image(data).pixels
    parts, meta = None, torch.zeros(4, dtype=torch.int64, device=_device_for_backend())
    if rank == 0:
        info = probe(container)
        parts = [split_band(container, t0, t1) if t1 > t0 else b"" for t0, t1 in band_tile_rows(info.height, info.tile_h, world)]
        meta = torch.tensor([info.width, info.height, info.channels, info.tile_h], dtype=torch.int64, device=meta.device)
    dist.broadcast(meta, src=0)
    w, h, c, _ = (int(x) for x in meta.tolist())
    mine = scatter_bytes(parts, src=0)
    px = decode_fn(mine) if len(mine) else np.zeros((0, w, c), np.uint8)
    rows = gather_bytes(np.ascontiguousarray(px).tobytes(), dst=0)
    if rank != 0:
        return None
    return np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(h, w, c)

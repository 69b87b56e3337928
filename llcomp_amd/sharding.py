"""Multi-GPU sharding of images over the ranks of a torch.distributed job: one process per GPU, backend "nccl" = RCCL
over xGMI (gloo + CPU staging exists for tests only).  SURVEY.md 8(e); the reference has no counterpart (single
process, single stream).

Slices have fresh state and slice-local borders, so a band of whole tile rows, encoded as an image of its own with
the same tiling, yields exactly the slices of the full image -- no halo exchange.  Work is dealt out in CHUNKS of tile
rows, round-robin over the ranks (several chunks per rank: the cost of a slice follows its entropy, not its pixels, so
fine interleaving balances non-uniform content); a rank stacks its chunks into one local image and codes it with one
device-resident codec object.

Everything stays in HBM.  Encode, per call (`images` images of one shape):
    1. every rank: llcomp_mi_codec_encode of its local image(s)         -> packed payload + slice lengths on its GPU
    2. all_gather of the slice-length tables (a few MB at most)
    3. ONE message per rank: its packed payload, GPU -> GPU, to the gathering rank (ncclSend / ncclRecv group)
    4. gathering rank: llcomp_mi_device_copy_segments interleaves the ranks' pieces chunk by chunk into image order
       behind the header and the permuted slice table                  -> one container per image, in HBM
Decode is the mirror image: broadcast of the table, segment copy into rank order, one message per rank, local decode;
the decoded bands stay on their ranks (gather_pixels() collects them when a single image is wanted).
The only host round trip is the handful of byte counts the send / recv sizes need.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import Codec, LlcompError, OK, _check, _lib

MAGIC_SLICED, HEADER = 0x9C, 24


def plan_chunks(height, tile_h, world, chunks_per_rank=4):
    """[(tile_row0, tile_row1, owner)]: consecutive chunks of whole tile rows, chunk i -> rank i % world.  Chunks are as
    even as the tile grid allows; with fewer tile rows than world * chunks_per_rank every chunk is one tile row."""
    tile_h = height if tile_h <= 0 or tile_h > height else tile_h
    nty = (height + tile_h - 1) // tile_h
    n_chunks = max(1, min(nty, world * max(1, chunks_per_rank)))
    out, t = [], 0
    for i in range(n_chunks):
        cnt = nty // n_chunks + (1 if i < nty % n_chunks else 0)
        out.append((t, t + cnt, i % world))
        t += cnt
    return out


def local_rows(height, tile_h, world, rank, chunks_per_rank=4):
    """[(y0, y1)] pixel-row ranges of the full image that `rank` codes, in the order it stacks them."""
    tile_h = height if tile_h <= 0 or tile_h > height else tile_h
    return [(t0 * tile_h, min(height, t1 * tile_h)) for t0, t1, o in plan_chunks(height, tile_h, world, chunks_per_rank) if o == rank]


class _HipBand:
    """The product's local coder: a device-resident codec object (llcomp_mi_codec_*) for `images` stacked bands."""

    def __init__(self, images, w, h, c, tile_w, tile_h, planar, device):
        self.codec = Codec(images, w, h, c, tile_w, tile_h, planar, device=device.index if device.index is not None else -1)
        self.n_slices = self.codec.n_slices
        self.device = device
        raw = images * w * h * c
        self.cap = min(self.codec.max_payload_bytes, 2 * raw + 64 * self.n_slices + 4096)
        self.payload = torch.empty(self.cap, dtype=torch.uint8, device=device)
        self.lens = torch.empty(self.n_slices, dtype=torch.int32, device=device)
        self.total = torch.zeros(1, dtype=torch.int64, device=device)
        self.status = torch.zeros(1, dtype=torch.int32, device=device)

    def encode(self, px):
        st = torch.cuda.current_stream(self.device).cuda_stream
        self.codec.encode(px.data_ptr(), self.payload.data_ptr(), self.cap, self.lens.data_ptr(), self.total.data_ptr(), self.status.data_ptr(), st)
        return self.payload, self.lens, self.total, self.status

    def decode(self, payload, payload_bytes, lens, out):
        st = torch.cuda.current_stream(self.device).cuda_stream
        self.codec.decode(payload.data_ptr(), payload_bytes, lens.data_ptr(), out.data_ptr(), self.status.data_ptr(), st)
        return self.status

    def check(self, status):
        rc = self.codec.status(int(status.item()))
        if rc != OK:
            raise LlcompError(rc)


def _copy_segments(src, dst, src_off, dst_off, lens, max_len):
    """dst[dst_off[i] : +lens[i]] = src[src_off[i] : +lens[i]] for every i, on the tensors' device."""
    n = int(lens.numel())
    if n == 0:
        return
    if src.is_cuda:
        st = torch.cuda.current_stream(src.device).cuda_stream
        for i in range(0, n, 65535):  # grid.y limit of one launch
            j = min(n, i + 65535)
            _check(_lib.load().llcomp_mi_device_copy_segments(src.data_ptr(), dst.data_ptr(), src_off[i:j].data_ptr(), dst_off[i:j].data_ptr(),
                                                               lens[i:j].data_ptr(), j - i, int(max_len), st))
    else:  # CPU tensors: the gloo test path
        for so, do, ln in zip(src_off.tolist(), dst_off.tolist(), lens.tolist()):
            dst[do:do + ln] = src[so:so + ln]


class ShardedCodec:
    """`images` images of shape (h, w, c), each sharded over all ranks of `group`; containers are gathered on rank `root`.

        sc = ShardedCodec(w, h, c, tile_w, tile_h, planar=True, images=B)
        band = sc.take_local(full)                 # this rank's rows of [B,h,w,c] (tests / bench; a real producer
                                                   #   delivers each rank only its rows)
        cont = sc.encode(band)                     # root: [(uint8 device tensor)] * B, others: None
        out = sc.decode(cont)                      # every rank: its decoded rows, device tensor [B, local_h, w, c]

    band_factory(images, w, local_h, c, tile_w, tile_h, planar, device) builds the local coder; the default is the HIP codec
    object, tests inject a CPU stand-in to exercise the distributed logic where no GPU exists."""

    def __init__(self, w, h, c, tile_w=0, tile_h=0, planar=True, images=1, group=None, root=0, chunks_per_rank=4, device=None,
                 band_factory=None):
        self.group = group
        self.rank, self.world, self.root = dist.get_rank(group), dist.get_world_size(group), root
        self.backend = dist.get_backend(group)
        self.w, self.h, self.c, self.images, self.planar = w, h, c, images, bool(planar)
        self.tile_w = w if tile_w <= 0 or tile_w > w else tile_w
        self.tile_h = h if tile_h <= 0 or tile_h > h else tile_h
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.device = device
        # tensors handed to the collectives: device memory with RCCL; gloo (tests) cannot send GPU tensors
        self.comm_device = device if self.backend == "nccl" else torch.device("cpu")
        self.chunks = plan_chunks(h, self.tile_h, self.world, chunks_per_rank)
        self.ntx = (w + self.tile_w - 1) // self.tile_w
        self.per_row = self.ntx * (c if self.planar else 1)  # slices per tile row
        self.nty = (h + self.tile_h - 1) // self.tile_h
        self.spf = self.per_row * self.nty                  # slices of one full image
        self.rows = [(t0 * self.tile_h, min(h, t1 * self.tile_h)) for t0, t1, o in self.chunks if o == self.rank]
        self.local_h = sum(y1 - y0 for y0, y1 in self.rows)
        self.local_slices = [sum((t1 - t0) for t0, t1, o in self.chunks if o == r) * self.per_row for r in range(self.world)]
        self.max_local = max(self.local_slices)
        factory = band_factory or _HipBand
        self.band = factory(images, w, self.local_h, c, self.tile_w, self.tile_h, self.planar, device) if self.local_h else None
        # slice permutation: container order (image, tile row, ...) <- rank-major order (rank, image, local tile row, ...)
        # position of (rank r, image b, local slice s) in the all_gather'ed [world, images * max_local] table
        src = np.empty((images, self.spf), dtype=np.int64)
        seen = [0] * self.world
        for t0, t1, o in self.chunks:
            n = (t1 - t0) * self.per_row
            for b in range(images):
                src[b, t0 * self.per_row:t1 * self.per_row] = o * images * self.max_local + b * self.local_slices[o] + seen[o] + np.arange(n)
            seen[o] += n
        self.perm = torch.from_numpy(src.reshape(-1)).to(device)
        # segments = (image, chunk): the unit the concatenator moves
        seg = [(b, ci) for b in range(images) for ci in range(len(self.chunks))]
        self.seg_first = torch.tensor([b * self.spf + self.chunks[ci][0] * self.per_row for b, ci in seg], dtype=torch.int64, device=device)
        self.seg_count = torch.tensor([(self.chunks[ci][1] - self.chunks[ci][0]) * self.per_row for b, ci in seg], dtype=torch.int64, device=device)
        self.seg_image = torch.tensor([b for b, ci in seg], dtype=torch.int64, device=device)
        self.header = torch.tensor(list(bytes([MAGIC_SLICED, 1, c, 1 if self.planar else 0]) + b"".join(
            int(v).to_bytes(4, "little") for v in (w, h, self.tile_w, self.tile_h, self.spf))), dtype=torch.uint8, device=device)

    # ---- helpers ------------------------------------------------------------------------------------------------
    def take_local(self, full):
        """this rank's rows of a full batch [images, h, w, c] (numpy or tensor) as one stacked device tensor"""
        t = torch.as_tensor(full)
        parts = [t[:, y0:y1] for y0, y1 in self.rows]
        if not parts:
            return torch.empty((self.images, 0, self.w, self.c), dtype=torch.uint8, device=self.device)
        return torch.cat(parts, dim=1).contiguous().to(self.device)

    def _to_comm(self, t):
        return t if t.device == self.comm_device else t.to(self.comm_device)

    def _all_lens(self, lens):
        """[world * images * max_local] int64 on self.device: every rank's slice lengths, zero padded"""
        mine = torch.zeros(self.images * self.max_local, dtype=torch.int32, device=self.device)
        if lens is not None:
            mine[: lens.numel()] = lens
        out = torch.empty(self.world * mine.numel(), dtype=torch.int32, device=self.comm_device)
        dist.all_gather_into_tensor(out, self._to_comm(mine), group=self.group)
        return out.to(self.device).to(torch.int64)

    def _segment_tables(self, lens_c):
        """from the container-order slice lengths [images * spf]: per (image, chunk) segment its byte count, its offset in
        the rank-major exchange buffers and its offset inside its image's payload"""
        csum = torch.zeros(lens_c.numel() + 1, dtype=torch.int64, device=self.device)
        torch.cumsum(lens_c, 0, out=csum[1:])
        seg_len = csum[self.seg_first + self.seg_count] - csum[self.seg_first]
        img_base = csum[self.seg_image * self.spf]
        in_image = csum[self.seg_first] - img_base  # offset of the segment inside its image's payload
        img_bytes = csum[torch.arange(1, self.images + 1, device=self.device) * self.spf] - csum[torch.arange(0, self.images, device=self.device) * self.spf]
        # rank-major order: (owner, image, chunk) = the order in which a rank's codec packs its local slices
        owner = torch.tensor([self.chunks[ci][2] for b in range(self.images) for ci in range(len(self.chunks))], dtype=torch.int64, device=self.device)
        order = torch.argsort(owner, stable=True)
        in_exchange = torch.empty_like(seg_len)
        in_exchange[order] = torch.cumsum(seg_len[order], 0) - seg_len[order]
        rank_bytes = torch.zeros(self.world, dtype=torch.int64, device=self.device).index_add_(0, owner, seg_len)
        return seg_len, in_exchange, in_image, img_bytes, rank_bytes

    # ---- encode -------------------------------------------------------------------------------------------------
    def encode(self, local_px):
        """local_px: uint8 device tensor [images, local_h, w, c] (this rank's stacked rows).  Returns on `root` a list of
        `images` uint8 device tensors, each a complete SLICED container; None on the other ranks."""
        payload = lens = status = None
        if self.band is not None:
            payload, lens, total, status = self.band.encode(local_px)
        all_lens = self._all_lens(lens)                      # collective 1: slice-length tables
        lens_c = all_lens[self.perm]                         # container order
        seg_len, in_exchange, in_image, img_bytes, rank_bytes = self._segment_tables(lens_c)
        host = torch.cat([rank_bytes, img_bytes]).cpu()      # the one host round trip: send / recv sizes
        if self.band is not None:
            self.band.check(status)
        rank_b, img_b = host[: self.world].tolist(), host[self.world:].tolist()
        starts = np.concatenate([[0], np.cumsum(rank_b)]).astype(np.int64)
        # collective 2: every rank's packed payload to the root, one message each, GPU to GPU
        if self.rank == self.root:
            exchange = torch.empty(int(starts[-1]) + 16, dtype=torch.uint8, device=self.comm_device)
            ops = [dist.P2POp(dist.irecv, exchange[starts[r]:starts[r + 1]], r, group=self.group)
                   for r in range(self.world) if r != self.root and rank_b[r]]
            for q in (dist.batch_isend_irecv(ops) if ops else []):
                q.wait()
            if rank_b[self.root]:
                exchange[starts[self.root]:starts[self.root + 1]] = self._to_comm(payload[: rank_b[self.root]])
            exchange = exchange.to(self.device)
        else:
            if rank_b[self.rank]:
                for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, self._to_comm(payload[: rank_b[self.rank]]), self.root, group=self.group)]):
                    q.wait()
            return None
        # root: containers = [header][table][payload], payload interleaved chunk by chunk by the device concatenator
        head = HEADER + 4 * self.spf
        sizes = [head + b for b in img_b]
        bases = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        buf = torch.empty(int(bases[-1]) + 16, dtype=torch.uint8, device=self.device)
        table = lens_c.to(torch.int32).view(self.images, self.spf)
        for b in range(self.images):
            buf[bases[b]:bases[b] + HEADER] = self.header
            buf[bases[b] + HEADER:bases[b] + head] = table[b].view(torch.uint8)  # little-endian u32, as on the wire
        dst_off = torch.as_tensor(bases[:-1], device=self.device)[self.seg_image] + head + in_image
        _copy_segments(exchange, buf, in_exchange, dst_off, seg_len, max(img_b) if img_b else 0)
        return [buf[bases[b]:bases[b + 1]] for b in range(self.images)]

    # ---- decode -------------------------------------------------------------------------------------------------
    def decode(self, containers):
        """containers: on `root` the list returned by encode (uint8 device tensors), None elsewhere.  Every rank returns
        its decoded rows [images, local_h, w, c] (uint8, on its device)."""
        head = HEADER + 4 * self.spf
        lens_c = torch.empty(self.images * self.spf, dtype=torch.int32, device=self.comm_device)
        if self.rank == self.root:
            for b, cont in enumerate(containers):
                if cont.numel() < head or bytes(cont[:HEADER].cpu().numpy()) != bytes(self.header.cpu().numpy()):
                    raise ValueError("container does not match this ShardedCodec's geometry")
                lens_c[b * self.spf:(b + 1) * self.spf] = self._to_comm(cont[HEADER:head].clone().view(torch.int32))  # (clone: dword alignment)
        dist.broadcast(lens_c, src=self.root, group=self.group)  # collective 1: the slice tables
        lens_c = lens_c.to(self.device).to(torch.int64)
        seg_len, in_exchange, in_image, img_bytes, rank_bytes = self._segment_tables(lens_c)
        rank_b = rank_bytes.cpu().tolist()                   # host round trip: message sizes
        starts = np.concatenate([[0], np.cumsum(rank_b)]).astype(np.int64)
        mine = None
        if self.rank == self.root:
            sizes = [int(c_.numel()) for c_ in containers]
            bases = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
            src = torch.cat(list(containers)) if len(containers) > 1 else containers[0]
            exchange = torch.empty(int(starts[-1]) + 16, dtype=torch.uint8, device=self.device)
            src_off = torch.as_tensor(bases[:-1], device=self.device)[self.seg_image] + head + in_image
            # the table may promise more than the payload holds (damaged container): clip, the decoder reports it
            seg_clip = torch.minimum(seg_len, torch.clamp(torch.as_tensor(bases[1:], device=self.device)[self.seg_image] - src_off, min=0))
            _copy_segments(src, exchange, src_off, in_exchange, seg_clip, int(max(sizes)))
            exchange = self._to_comm(exchange)
            ops = [dist.P2POp(dist.isend, exchange[starts[r]:starts[r + 1]], r, group=self.group)
                   for r in range(self.world) if r != self.root and rank_b[r]]
            for q in (dist.batch_isend_irecv(ops) if ops else []):  # collective 2: one message per rank
                q.wait()
            if rank_b[self.root]:
                mine = exchange[starts[self.root]:starts[self.root + 1]].to(self.device)
        elif rank_b[self.rank]:
            mine = torch.empty(rank_b[self.rank] + 16, dtype=torch.uint8, device=self.comm_device)
            for q in dist.batch_isend_irecv([dist.P2POp(dist.irecv, mine[: rank_b[self.rank]], self.root, group=self.group)]):
                q.wait()
            mine = mine.to(self.device)
        out = torch.empty((self.images, self.local_h, self.w, self.c), dtype=torch.uint8, device=self.device)
        if self.band is None:
            return out
        # this rank's slice lengths in ITS codec's order (image, local tile rows): the inverse of the permutation
        all_order = torch.empty(self.world * self.images * self.max_local, dtype=torch.int64, device=self.device)
        all_order[self.perm] = lens_c
        per = self.local_slices[self.rank]
        base = self.rank * self.images * self.max_local
        my_lens = all_order[base: base + self.images * per].to(torch.int32).contiguous()
        if mine is None:
            mine = torch.zeros(16, dtype=torch.uint8, device=self.device)
        status = self.band.decode(mine, rank_b[self.rank], my_lens, out)
        self.band.check(status)
        return out

    def gather_pixels(self, local_out):
        """root: the full batch [images, h, w, c] assembled from every rank's decoded rows (one message per rank); others None"""
        flat = self._to_comm(local_out.contiguous().view(-1))
        row_bytes = self.w * self.c
        heights = [sum(min(self.h, t1 * self.tile_h) - t0 * self.tile_h for t0, t1, o in self.chunks if o == r) for r in range(self.world)]
        if self.rank != self.root:
            if flat.numel():
                for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, flat, self.root, group=self.group)]):
                    q.wait()
            return None
        parts, ops = [], []
        for r in range(self.world):
            buf = flat if r == self.root else torch.empty(self.images * heights[r] * row_bytes, dtype=torch.uint8, device=self.comm_device)
            if r != self.root and buf.numel():
                ops.append(dist.P2POp(dist.irecv, buf, r, group=self.group))
            parts.append(buf)
        for q in (dist.batch_isend_irecv(ops) if ops else []):
            q.wait()
        full = torch.empty((self.images, self.h, self.w, self.c), dtype=torch.uint8, device=self.device)
        seen = [0] * self.world
        for t0, t1, o in self.chunks:
            y0, y1 = t0 * self.tile_h, min(self.h, t1 * self.tile_h)
            band = parts[o].view(self.images, heights[o], self.w, self.c)[:, seen[o]:seen[o] + (y1 - y0)]
            full[:, y0:y1] = band.to(self.device)
            seen[o] += y1 - y0
        return full

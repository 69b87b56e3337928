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
    3. ONE variable-size all-to-all of the packed payloads, GPU -> GPU (RCCL alltoallv over xGMI): image b is gathered on
       rank b % world, so all 7 links of every GPU carry bytes in both directions instead of one GPU's inbound links
       carrying the whole bitstream (root=r funnels everything to rank r when one rank must hold all containers)
    4. gathering ranks: llcomp_mi_device_copy_segments interleaves the ranks' pieces chunk by chunk into image order
       behind the header and the permuted slice table                  -> complete containers, in HBM
Decode is the mirror image: all_gather of the tables, segment copy into exchange order, all-to-all, local decode; the
decoded bands stay on their ranks (gather_pixels() collects them when a single image is wanted).
The only host round trip per direction is the world x world matrix of message sizes.
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

from . import Codec, LlcompError, OK, _check, _lib

MAGIC_SLICED, HEADER = 0x9C, 24


def plan_chunks(height, tile_h, world, chunks_per_rank=4):
    """[(tile_row0, tile_row1, owner)]: consecutive chunks of whole tile rows, chunk i -> rank i % world.  Chunks are as
    even as the tile grid allows; with fewer tile rows than world * chunks_per_rank every chunk is one tile row.
    ONE implementation for every multi-GPU path: llcomp_mi_plan_chunks (csrc/container.cpp) -- the in-process device lists of
    the C ABI split an image exactly like the ranks here."""
    from . import plan_chunks as _plan

    return _plan(height, tile_h if tile_h > 0 else 0, world, chunks_per_rank)


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

    def status_code(self, bits):
        return self.codec.status(int(bits))

    def check(self, status):
        rc = self.status_code(int(status.item()))
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
    """`images` images of shape (h, w, c), each sharded over all ranks of `group`.

        sc = ShardedCodec(w, h, c, tile_w, tile_h, planar=True, images=B)          # containers spread over the ranks
        band = sc.take_local(full)       # this rank's rows of [B,h,w,c], stacked (tests / bench; a real producer delivers
                                         #   each rank only its rows); frame f of the stack is image sc.frame_images[f]
        conts = sc.encode(band)          # {image index: uint8 device tensor = complete SLICED container} for the images
                                         #   this rank gathers (sc.my_images)
        out = sc.decode(conts)           # every rank: its decoded rows, device tensor [B, local_h, w, c]

    Where a container is assembled: root=None (default) spreads the images round-robin over the ranks (image b is gathered
    on rank b % world), so the exchange is an all-to-all that uses every xGMI link in both directions; root=r gathers every
    image on rank r (a funnel: that rank's seven inbound links carry the whole bitstream).

    band_factory(images, w, local_h, c, tile_w, tile_h, planar, device) builds the local coder; the default is the HIP codec
    object, tests inject a CPU stand-in to exercise the distributed logic where no GPU exists."""

    def __init__(self, w, h, c, tile_w=0, tile_h=0, planar=True, images=1, group=None, root=None, chunks_per_rank=4, device=None,
                 band_factory=None, force_exchange=None):
        self.group = group
        # test hook: run the payload collective even where it is the identity (world 1), so that the RCCL alltoallv on
        # device tensors executes on a one-GPU box (LLCOMP_MI_FORCE_EXCHANGE=1 does the same for bench.py)
        self.force_exchange = bool(int(os.environ.get("LLCOMP_MI_FORCE_EXCHANGE", "0"))) if force_exchange is None else bool(force_exchange)
        self.exchanges = 0  # payload collectives actually executed (tests assert on it)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.w, self.h, self.c, self.images, self.planar = w, h, c, images, bool(planar)
        self.tile_w = w if tile_w <= 0 or tile_w > w else tile_w
        self.tile_h = h if tile_h <= 0 or tile_h > h else tile_h
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.device = device
        # tensors handed to the collectives: device memory with RCCL; gloo (tests) cannot move GPU tensors
        self.comm_device = device if self.backend == "nccl" else torch.device("cpu")
        world = self.world
        self.root_of = [(b % world) if root is None else int(root) for b in range(images)]
        self.my_images = [b for b in range(images) if self.root_of[b] == self.rank]
        # a rank's codec packs its frames back to back: frames ordered by (gathering rank, image) make the bytes for one
        # destination contiguous
        self.frame_images = sorted(range(images), key=lambda b: (self.root_of[b], b))
        frame_of = {b: f for f, b in enumerate(self.frame_images)}
        self.chunks = plan_chunks(h, self.tile_h, world, chunks_per_rank)
        self.ntx = (w + self.tile_w - 1) // self.tile_w
        self.per_row = self.ntx * (c if self.planar else 1)  # slices per tile row
        # no slice stream is longer than this (geometry.hpp: 13 bytes per sample + slack); lengths read from a container
        # are clamped to it before anything is sized or copied by them
        # (with geometry.hpp's ceiling: it goes to the device as a u32, and make_geometry caps a slice's capacity the same way)
        self.slice_cap = min((13 * self.tile_w * self.tile_h * (1 if self.planar else c) + 32 + 15) // 16 * 16, 0x7FFFFFF0)
        self.nty = (h + self.tile_h - 1) // self.tile_h
        self.spf = self.per_row * self.nty                  # slices of one full image
        self.rows = [(t0 * self.tile_h, min(h, t1 * self.tile_h)) for t0, t1, o in self.chunks if o == self.rank]
        self.local_h = sum(y1 - y0 for y0, y1 in self.rows)
        self.local_slices = [sum((t1 - t0) for t0, t1, o in self.chunks if o == r) * self.per_row for r in range(world)]
        self.max_local = max(self.local_slices)
        factory = band_factory or _HipBand
        self.band = factory(images, w, self.local_h, c, self.tile_w, self.tile_h, self.planar, device) if self.local_h else None
        # Everything below is geometry, fixed at construction.  The slice lengths of every rank travel in ONE all_gather'ed
        # int32 table [world][images * max_local + 1] (the last word of a row = that rank's coder status); `n1` = its row stride.
        n_ch = len(self.chunks)
        self.n1 = n1 = images * self.max_local + 1
        # slice permutation: container order (image, tile row, ...) <- position in the flattened gathered table
        src = np.empty((images, self.spf), dtype=np.int64)
        enc_start = np.empty((images, n_ch), dtype=np.int64)   # where the slices of segment (image, chunk) start in that table
        seen = [0] * world
        for ci, (t0, t1, o) in enumerate(self.chunks):
            n = (t1 - t0) * self.per_row
            for b in range(images):
                first = o * n1 + frame_of[b] * self.local_slices[o] + seen[o]
                src[b, t0 * self.per_row:t1 * self.per_row] = first + np.arange(n)
                enc_start[b, ci] = first
            seen[o] += n
        # segments = (image, chunk), index b * n_ch + ci: the unit the concatenator moves, a contiguous run of slices both in
        # the owner's table and in the container
        self.n_seg = images * n_ch
        seg = [(b, ci) for b in range(images) for ci in range(n_ch)]
        self.seg_count_h = np.array([(self.chunks[ci][1] - self.chunks[ci][0]) * self.per_row for b, ci in seg], dtype=np.int64)
        self.seg_image_h = np.array([b for b, ci in seg], dtype=np.int64)
        self.seg_owner_h = np.array([self.chunks[ci][2] for b, ci in seg], dtype=np.int64)
        self.seg_root_h = np.array([self.root_of[b] for b, ci in seg], dtype=np.int64)
        dec_start = np.array([b * self.spf + self.chunks[ci][0] * self.per_row for b, ci in seg], dtype=np.int64)
        # exchange order: (gathering rank, coding rank, frame, chunk) -- the order of the bytes in a gathering rank's receive
        # buffer (encode) / send buffer (decode)
        key = [(self.root_of[b], self.chunks[ci][2], frame_of[b], ci) for b, ci in seg]
        self.xorder_h = np.array(sorted(range(len(seg)), key=lambda i: key[i]), dtype=np.int64)
        self.seg_mine_h = np.array([i for i, (b, ci) in enumerate(seg) if self.root_of[b] == self.rank], dtype=np.int64)
        t = lambda v: torch.as_tensor(np.ascontiguousarray(v), dtype=torch.int64).to(device)  # noqa: E731
        self.enc_start_d, self.dec_start_d, self.seg_count_d = t(enc_start.reshape(-1)), t(dec_start), t(self.seg_count_h)
        # container-order slice tables of the images I gather, as indices into the flattened gathered table
        self.perm_mine_d = t(src[self.my_images].reshape(-1)) if self.my_images else None
        # decode: the tables of all images arrive as [world][per_root * spf + 1] (a flag word per rank); container order of
        # image b = row root_of[b], slot = its position among that rank's images
        self.per_root = max(1, max(sum(1 for b in range(images) if self.root_of[b] == r) for r in range(world)))
        slot, pick = [0] * world, []
        for b in range(images):
            pick.append(self.root_of[b] * (self.per_root * self.spf + 1) + slot[self.root_of[b]] * self.spf)
            slot[self.root_of[b]] += 1
        self.dec_src_d = t((np.array(pick, dtype=np.int64)[:, None] + np.arange(self.spf)[None, :]).reshape(-1))
        # ... and this rank's slices in ITS codec's order (frame, local tile rows) as indices into that container-order table
        inv = np.zeros(images * self.local_slices[self.rank], dtype=np.int64)
        mine_rows = src - self.rank * n1
        own = (src >= self.rank * n1) & (src < self.rank * n1 + n1 - 1)
        bb, ss = np.nonzero(own)
        inv[mine_rows[bb, ss]] = bb * self.spf + ss
        self.inv_mine_d = t(inv)
        self._pending = None
        self._pending_decode = None
        self.header = torch.tensor(list(bytes([MAGIC_SLICED, 1, c, 1 if self.planar else 0]) + b"".join(
            int(v).to_bytes(4, "little") for v in (w, h, self.tile_w, self.tile_h, self.spf))), dtype=torch.uint8, device=device)

    # ---- helpers ------------------------------------------------------------------------------------------------
    def take_local(self, full):
        """this rank's rows of a full batch [images, h, w, c] (numpy or tensor), frames in codec order (frame_images), as one
        stacked tensor on this rank's device"""
        t = torch.as_tensor(full)[self.frame_images]
        parts = [t[:, y0:y1] for y0, y1 in self.rows]
        if not parts:
            return torch.empty((self.images, 0, self.w, self.c), dtype=torch.uint8, device=self.device)
        return torch.cat(parts, dim=1).contiguous().to(self.device)

    def _to_comm(self, t):
        return t if t.device == self.comm_device else t.to(self.comm_device)

    def _gather_lens(self, lens, status=None):
        """int32 [world * n1] on self.device: every rank's slice lengths (zero padded) followed by its coder status word.  The
        status rides along in the same all_gather, so every rank learns of a failed local encode before the payload collective
        and all of them raise together (nobody is left waiting in all_to_all)."""
        mine = torch.zeros(self.n1, dtype=torch.int32, device=self.device)
        if lens is not None:
            mine[: lens.numel()] = lens
        if status is not None:
            mine[self.n1 - 1] = status.reshape(-1)[0].to(self.device)
        out = torch.empty(self.world * self.n1, dtype=torch.int32, device=self.comm_device)
        dist.all_gather_into_tensor(out, self._to_comm(mine), group=self.group)
        return out.to(self.device)

    def _as_lengths(self, t):
        """int32 words read as the u32 they are on the wire, clamped to what a slice can hold"""
        return torch.clamp(t.to(torch.int64) & 0xFFFFFFFF, max=self.slice_cap)

    def _range_sums(self, table_i32, start_d):
        """int64 [n_seg] on self.device: bytes of every (image, chunk) segment = sum of its slices' lengths (read as u32, clamped
        to a slice's capacity) -- ONE launch (llcomp_mi_device_range_sums); torch arithmetic for the CPU tensors of the gloo tests"""
        if table_i32.is_cuda:
            out = torch.empty(self.n_seg, dtype=torch.int64, device=self.device)
            st = torch.cuda.current_stream(self.device).cuda_stream
            _check(_lib.load().llcomp_mi_device_range_sums(table_i32.data_ptr(), start_d.data_ptr(), self.seg_count_d.data_ptr(), out.data_ptr(),
                                                           self.n_seg, self.slice_cap, st))
            return out
        csum = torch.zeros(table_i32.numel() + 1, dtype=torch.int64)
        torch.cumsum(self._as_lengths(table_i32), 0, out=csum[1:])
        return csum[start_d + self.seg_count_d] - csum[start_d]

    def _host_tables(self, seg_len):
        """From the byte count of every (image, chunk) segment (numpy int64 [n_seg], on the host: a few hundred numbers): its
        offset in the gathering rank's exchange buffer, its offset inside its image's payload, every image's payload bytes and
        the message matrix M[coding rank][gathering rank]."""
        per_image = seg_len.reshape(self.images, -1)
        in_image = (np.cumsum(per_image, axis=1) - per_image).reshape(-1)
        img_bytes = per_image.sum(axis=1)
        M = np.zeros((self.world, self.world), dtype=np.int64)
        np.add.at(M, (self.seg_owner_h, self.seg_root_h), seg_len)
        ordered = seg_len[self.xorder_h]
        run = np.cumsum(ordered) - ordered                               # offset in the concatenation of ALL exchange buffers
        recv_total = M.sum(axis=0)                                       # bytes every gathering rank holds
        base = np.cumsum(recv_total) - recv_total
        in_exchange = np.empty_like(seg_len)
        in_exchange[self.xorder_h] = run - base[self.seg_root_h[self.xorder_h]]   # offset inside ITS gathering rank's buffer
        return in_exchange, in_image, img_bytes, M

    def _offsets_to_device(self, *rows):
        """small int64 rows (numpy) -> one [k, n] tensor on self.device in one asynchronous copy from pinned memory (no host wait
        behind the work already queued on the stream)"""
        t = torch.from_numpy(np.ascontiguousarray(np.stack(rows).astype(np.int64)))
        if self.device.type == "cuda":
            return t.pin_memory().to(self.device, non_blocking=True)
        return t

    def _exchange(self, send, send_split, recv_split, m_max):
        """variable-size all-to-all of bytes (RCCL alltoallv on device tensors); returns the receive buffer on self.device"""
        n_send = int(sum(send_split))
        if self.world == 1 and not self.force_exchange:
            return send  # one rank: what it would send to itself is already in place
        self.exchanges += 1
        send_split, recv_split = [int(x) for x in send_split], [int(x) for x in recv_split]
        recv = torch.empty(sum(recv_split) + 16, dtype=torch.uint8, device=self.comm_device)
        src = self._to_comm(send[:n_send]).contiguous() if n_send else torch.empty(0, dtype=torch.uint8, device=self.comm_device)
        if m_max <= self.MAX_MESSAGE:  # (m_max = the largest message of ANY rank pair: the same decision on every rank)
            dist.all_to_all_single(recv[: sum(recv_split)], src, output_split_sizes=recv_split, input_split_sizes=send_split, group=self.group)
            return recv.to(self.device)
        # A message beyond MAX_MESSAGE bytes goes in rounds of at most that many bytes per peer: a self-message above 1 GiB
        # arrives with only its first half written (RCCL 2.26.6, all_to_all_single and send/recv alike, deterministic:
        # profiles/r04_rccl_self_message.txt); real multi-rank messages are far smaller.  Every rank derives the same number of
        # rounds from the same message matrix, so the collectives stay matched.
        rounds = -(-int(m_max) // self.MAX_MESSAGE)
        s_off = np.concatenate([[0], np.cumsum(send_split)]).astype(np.int64)
        r_off = np.concatenate([[0], np.cumsum(recv_split)]).astype(np.int64)
        for k in range(rounds):
            lo = k * self.MAX_MESSAGE
            s_len = [max(0, min(self.MAX_MESSAGE, n - lo)) for n in send_split]
            r_len = [max(0, min(self.MAX_MESSAGE, n - lo)) for n in recv_split]
            s_buf = torch.cat([src[s_off[p] + lo: s_off[p] + lo + s_len[p]] for p in range(self.world)]) if sum(s_len) else src[:0]
            r_buf = torch.empty(sum(r_len), dtype=torch.uint8, device=self.comm_device)
            dist.all_to_all_single(r_buf, s_buf, output_split_sizes=r_len, input_split_sizes=s_len, group=self.group)
            at = 0
            for p in range(self.world):
                recv[r_off[p] + lo: r_off[p] + lo + r_len[p]] = r_buf[at: at + r_len[p]]
                at += r_len[p]
        return recv.to(self.device)

    MAX_MESSAGE = 1 << 30

    # ---- encode -------------------------------------------------------------------------------------------------
    def encode(self, local_px):
        """local_px: uint8 device tensor [images, local_h, w, c] (this rank's stacked rows, frames in frame_images order).
        Returns {image index: uint8 device tensor holding the complete SLICED container} for the images this rank gathers.
        = encode_begin + encode_finish; two ShardedCodec objects on two HIP streams can interleave the halves so that the
        exchange of one batch overlaps the coding of the other (bench.py does)."""
        self.encode_begin(local_px)
        return self.encode_finish()

    def encode_begin(self, local_px):
        """enqueue the local coding on the current stream (asynchronous, no collective, no host wait)"""
        self._pending = self.band.encode(local_px) if self.band is not None else None

    def encode_finish(self):
        """the exchange: slice-table all_gather, segment byte counts to the host, all-to-all of the payloads, concatenator.
        Device work beside the collectives: one range-sum launch, one gather of my images' slice tables, one concatenator
        launch (round 2 ran ~60 small torch kernels here, each queued behind the other streams' slice kernels)."""
        payload = lens = status = None
        if self._pending is not None:
            payload, lens, total, status = self._pending
        self._pending = None
        gathered = self._gather_lens(lens, status)           # collective 1: slice-length tables (+ every rank's status word)
        seg_len_d = self._range_sums(gathered, self.enc_start_d)
        status_d = gathered.view(self.world, self.n1)[:, self.n1 - 1].to(torch.int64)
        host = torch.cat([seg_len_d, status_d]).cpu().numpy()  # the one host round trip: segment bytes (=> every size) + verdicts
        seg_len = host[: self.n_seg]
        for r, bits in enumerate(host[self.n_seg:].tolist()):  # the same verdict on every rank, before collective 2
            if bits:
                code = self.band.status_code(bits) if hasattr(self.band, "status_code") else 7
                raise LlcompError(code, f"rank {r}'s local encode failed, status bits {bits}")
        in_exchange, in_image, img_bytes, M = self._host_tables(seg_len)
        if payload is None:
            payload = torch.empty(0, dtype=torch.uint8, device=self.device)
        # collective 2: every coding rank's packed payload to the gathering ranks, GPU to GPU
        exchange = self._exchange(payload, M[self.rank].tolist(), M[:, self.rank].tolist(), int(M.max()))
        if not self.my_images:
            return {}
        # containers of my images = [header][table][payload], payload interleaved chunk by chunk by the device concatenator
        head = HEADER + 4 * self.spf
        sizes = [head + int(img_bytes[b]) for b in self.my_images]
        bases = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        buf = torch.empty(int(bases[-1]) + 16, dtype=torch.uint8, device=self.device)
        table = self._as_lengths(gathered[self.perm_mine_d]).to(torch.int32).view(len(self.my_images), self.spf)  # container order
        base_of = np.zeros(self.images, dtype=np.int64)
        for j, b in enumerate(self.my_images):
            buf[bases[j]:bases[j] + HEADER] = self.header
            buf[bases[j] + HEADER:bases[j] + head] = table[j].view(torch.uint8)  # little-endian u32, as on the wire
            base_of[b] = bases[j]
        mine = self.seg_mine_h
        offs = self._offsets_to_device(in_exchange[mine], base_of[self.seg_image_h[mine]] + head + in_image[mine], seg_len[mine])
        _copy_segments(exchange, buf, offs[0], offs[1], offs[2], max(sizes))
        if offs.is_cuda:
            offs.record_stream(torch.cuda.current_stream(self.device))
        return {b: buf[bases[j]:bases[j + 1]] for j, b in enumerate(self.my_images)}

    # ---- decode -------------------------------------------------------------------------------------------------
    def decode(self, containers, validate=True):
        """containers: {image index: uint8 device tensor} for the images this rank holds (what encode returned).  Every rank
        returns its decoded rows [images, local_h, w, c] (uint8, on its device; frames in frame_images order).
        validate=False skips the comparison of the 24 header bytes with this object's geometry (a host round trip per
        container) -- for containers that come straight from encode().
        = decode_begin + decode_finish; several ShardedCodec objects on their own HIP streams queue all their decodes before
        anyone waits for a status (bench.py does)."""
        self.decode_begin(containers, validate)
        return self.decode_finish()

    def decode_finish(self):
        """wait for the decode queued by decode_begin (on the stream it was queued on), raise on a damaged stream, return
        this rank's rows"""
        out, status = self._pending_decode[:2]
        self._pending_decode = None
        if status is not None:
            self.band.check(status)
        return out

    def decode_begin(self, containers, validate=True):
        """the exchange back (slice-table all_gather, segment byte counts to the host, all-to-all of the payloads) and the local
        decode, enqueued on the current stream; nothing waits for the decoded pixels"""
        head = HEADER + 4 * self.spf
        # [per_root * spf slice lengths | 1 flag]: a rank that holds a container it cannot use says so in the flag word of the
        # same all_gather, and every rank raises together (a lone raise would leave the others waiting in the collective)
        row = self.per_root * self.spf + 1
        mine_tab = torch.zeros(row, dtype=torch.int32, device=self.comm_device)
        for j, b in enumerate(self.my_images):
            cont = containers.get(b) if hasattr(containers, "get") else containers[b]
            if cont is None or cont.numel() < head or (validate and bytes(cont[:HEADER].cpu().numpy()) != bytes(self.header.cpu().numpy())):
                mine_tab[-1] = 1
                continue
            mine_tab[j * self.spf:(j + 1) * self.spf] = self._to_comm(cont[HEADER:head].clone().view(torch.int32))  # (clone: dword alignment)
        tabs = torch.empty(self.world * row, dtype=torch.int32, device=self.comm_device)
        dist.all_gather_into_tensor(tabs, mine_tab, group=self.group)    # collective 1: the slice tables of every image
        tabs = tabs.to(self.device)
        # The tables come out of containers: read as the u32 they are and clamped to a slice's capacity wherever a size is
        # derived from them, so that a damaged table can neither go negative nor size a copy beyond what the geometry allows
        # (the decoder reports the damage).
        lens_raw = tabs[self.dec_src_d]                                  # container order, [images * spf] int32
        seg_len_d = self._range_sums(lens_raw, self.dec_start_d)
        flags_d = tabs.view(self.world, row)[:, row - 1].to(torch.int64)
        host = torch.cat([seg_len_d, flags_d]).cpu().numpy()             # the one host round trip: segment bytes + every rank's verdict on its containers
        bad = host[self.n_seg:].tolist()
        if any(bad):
            raise ValueError("container does not match this ShardedCodec's geometry (rank(s) %s)" % [r for r, f in enumerate(bad) if f])
        seg_len = host[: self.n_seg]
        in_exchange, in_image, img_bytes, M = self._host_tables(seg_len)
        # my containers -> send buffer ordered (coding rank, frame, chunk); the table may promise more than a damaged
        # container holds: clip, the decoder reports it
        n_out = int(M[:, self.rank].sum())
        send = torch.zeros(n_out + 16, dtype=torch.uint8, device=self.device)
        n_ch = len(self.chunks)
        for b in self.my_images:  # one concatenator launch per container: its (image, chunk) segments are consecutive
            cont = containers[b]
            idx = np.arange(b * n_ch, (b + 1) * n_ch)
            src_off = head + in_image[idx]
            seg_clip = np.minimum(seg_len[idx], np.maximum(int(cont.numel()) - src_off, 0))
            offs = self._offsets_to_device(src_off, in_exchange[idx], seg_clip)
            _copy_segments(cont, send, offs[0], offs[1], offs[2], int(cont.numel()))
            if offs.is_cuda:
                offs.record_stream(torch.cuda.current_stream(self.device))
        # collective 2: every rank gets the bytes of its slices, already in its codec's order
        recv = self._exchange(send, M[:, self.rank].tolist(), M[self.rank].tolist(), int(M.max()))
        out = torch.empty((self.images, self.local_h, self.w, self.c), dtype=torch.uint8, device=self.device)
        if self.band is None:
            self._pending_decode = (out, None)
            return
        # this rank's slice lengths in ITS codec's order (frame, local tile rows)
        my_lens = self._as_lengths(lens_raw[self.inv_mine_d]).to(torch.int32).contiguous()
        # (recv and my_lens stay referenced until decode_finish: the kernels that read them are only queued here)
        self._pending_decode = (out, self.band.decode(recv, int(M[self.rank].sum()), my_lens, out), recv, my_lens)

    def gather_pixels(self, local_out, dst=0):
        """rank `dst`: the full batch [images, h, w, c] assembled from every rank's decoded rows (one message per rank);
        None elsewhere"""
        flat = self._to_comm(local_out.contiguous().view(-1))
        row_bytes = self.w * self.c
        heights = [sum(min(self.h, t1 * self.tile_h) - t0 * self.tile_h for t0, t1, o in self.chunks if o == r) for r in range(self.world)]
        if self.rank != dst:
            if flat.numel():
                for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, flat, dst, group=self.group)]):
                    q.wait()
            return None
        parts, ops = [], []
        for r in range(self.world):
            buf = flat if r == dst else torch.empty(self.images * heights[r] * row_bytes, dtype=torch.uint8, device=self.comm_device)
            if r != dst and buf.numel():
                ops.append(dist.P2POp(dist.irecv, buf, r, group=self.group))
            parts.append(buf)
        for q in (dist.batch_isend_irecv(ops) if ops else []):
            q.wait()
        full = torch.empty((self.images, self.h, self.w, self.c), dtype=torch.uint8, device=self.device)
        seen = [0] * self.world
        inv = torch.as_tensor(self.frame_images, device=self.device)
        for t0, t1, o in self.chunks:
            y0, y1 = t0 * self.tile_h, min(self.h, t1 * self.tile_h)
            band = parts[o].view(self.images, heights[o], self.w, self.c)[:, seen[o]:seen[o] + (y1 - y0)]
            full[inv, y0:y1] = band.to(self.device)  # frame f of the stack is image frame_images[f]
            seen[o] += y1 - y0
        return full

/* llcomp_mi.h -- C ABI of the MI355X-native llcomp coding path (libllcomp_mi.so).
 *
 * The reference (vovach777/llcomp) is a header-only C++ library with exactly two entry points and no
 * FFI of its own; this header is the boundary a maintainer would bind instead of them:
 *
 *   llcomp::compressImage(const std::vector<uint8_t>& rgb, int w, int h, int channels)
 *        -> std::vector<uint8_t>                         /root/reference/llcomp.hpp:358
 *   llcomp::decompressImage(const std::vector<uint8_t>& data) -> RawImage{pixels,width,height,channels}
 *                                                        /root/reference/llcomp.hpp:454-461
 *   callers: llcompc.cpp:33, llcompd.cpp:26
 *
 * Everything here is plain C: pointers, sizes, integer status codes.  No torch, no C++ types.
 * All compute runs in hand-written HIP kernels for gfx950; there is NO CPU code path behind these
 * calls -- without a HIP device they return LLCOMP_MI_NO_DEVICE.
 * include/llcomp_mi.hpp layers the reference's own C++ signatures on top; INTEGRATION.md shows the
 * binding.
 */
#ifndef LLCOMP_MI_H
#define LLCOMP_MI_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI 4 (round 6: device lists -- llcomp_mi_opts.devices, llcomp_mi_decode_devices, llcomp_mi_stream_create_multi,
 * llcomp_mi_plan_chunks, llcomp_mi_codec_get_counters).  The library and its callers are built from ONE header: structs have one layout per ABI version (llcomp_mi_opts is
 * checked through struct_size and refused when it differs; llcomp_mi_info and llcomp_mi_stream_result are written in full),
 * so a binding compares llcomp_mi_abi_version() with the LLCOMP_MI_ABI_VERSION it was generated from and refuses to run on
 * a mismatch -- there is no cross-version compatibility mode. */
#define LLCOMP_MI_ABI_VERSION 4

/* Wire formats.  LEGACY is the reference's own: [0x79][channels u8][width u16 LE][height u16 LE] + ONE
 * range-coded stream (llcomp.hpp:375-378); it is a single serial chain (one GPU lane).  SLICED is this
 * project's container: independent slices, each a bare reference-compatible stream with fresh state:
 *   [0x9C][ver=1][channels][flags bit0=planar, bit1=small model] [w u32][h u32][tile_w u32][tile_h u32][n_slices u32]
 *   [len u32] x n_slices  [payload bytes ...]                                  (all little-endian)
 * "Small model" = the bitstream of a reference built with `LargeModel = false` (llcomp.hpp:21, 26-32, 427-429: the two
 * quant5 terms are left out of the context).  The reference's header does not record that build option (the magic
 * stays 0x79), so for the LEGACY format the caller has to say so on both sides; the SLICED header carries a flag. */
#define LLCOMP_MI_MAGIC_LEGACY 0x79
#define LLCOMP_MI_MAGIC_SLICED 0x9C
#define LLCOMP_MI_SLICED_HEADER_BYTES 24

typedef enum llcomp_mi_status {
    LLCOMP_MI_OK = 0,
    LLCOMP_MI_BAD_MAGIC = 1,       /* reference throws "Invalid magic number"  (llcomp.hpp:465-467) */
    LLCOMP_MI_BAD_EXPONENT = 2,    /* reference throws "Invalid exponent"      (llcomp.hpp:232-234) */
    LLCOMP_MI_TRUNCATED = 3,       /* header or slice table longer than the data (reference: UB, D5) */
    LLCOMP_MI_BAD_ARGS = 4,        /* null pointers, zero sizes, unsupported channel count, bad opts */
    LLCOMP_MI_OUT_OF_RANGE = 5,    /* legacy format with w or h > 65535, or w*h*c >= 2^31 (reference: silent truncation, D4) */
    LLCOMP_MI_OUTPUT_OVERFLOW = 6, /* caller-provided output capacity too small (reference: heap overflow, D1) */
    LLCOMP_MI_HIP_ERROR = 7,       /* a HIP call failed, or a kernel found its own launch assumptions violated and refused to run */
    LLCOMP_MI_NO_DEVICE = 8,
    LLCOMP_MI_NOMEM = 9,
    LLCOMP_MI_BUSY = 10,           /* streaming pipeline: every slot is occupied / the oldest job is still in flight */
    LLCOMP_MI_DEVICE_FAILED = 11   /* a call over a device list: one of the devices failed (HIP error, out of memory, no such device) and
                                      nothing was published; llcomp_mi_last_device_error tells which one and why.  Verdicts about the
                                      DATA (BAD_EXPONENT, TRUNCATED, OUTPUT_OVERFLOW) come back as themselves from any device, and so
                                      does NO_DEVICE (a machine without any HIP device: no member of the list to blame). */
} llcomp_mi_status;

typedef enum llcomp_mi_format { LLCOMP_MI_FORMAT_LEGACY = 0, LLCOMP_MI_FORMAT_SLICED = 1 } llcomp_mi_format;

typedef struct llcomp_mi_opts {
    uint32_t struct_size; /* = sizeof(llcomp_mi_opts) */
    uint32_t format;      /* llcomp_mi_format */
    uint32_t tile_w;      /* slice width  in pixels, 0 = full width   (SLICED only) */
    uint32_t tile_h;      /* slice height in pixels, 0 = full height  (SLICED only) */
    uint32_t planar;      /* 1 = one slice per colour-transformed channel plane, 0 = channels interleaved */
    int32_t device;       /* HIP device ordinal, -1 = current device */
    uint32_t small_model; /* 1 = code like a reference built with LargeModel = false */
    /* One image over several GPUs, in this process (BASELINE config 4; SURVEY 8b "device list").  n_devices == 0: one device
     * (`device`).  n_devices >= 1 (SLICED only; at most LLCOMP_MI_MAX_DEVICES): the image's tile rows are dealt in chunks
     * round-robin over devices[0..n_devices) (llcomp_mi_plan_chunks), every device gets ONLY its rows over its own PCIe link,
     * codes them with its own lane and copies its payload straight to its place in the container -- there is no exchange between
     * the GPUs and no collective, and the container is byte-identical to the one-device container.  An ordinal may repeat (two
     * lanes on one GPU: how the path is tested on a one-GPU box).  A LEGACY stream is one serial chain and does not shard:
     * devices[0] codes it.  `device` is ignored when n_devices > 0. */
    uint32_t n_devices;
    const int32_t* devices;
    uint32_t chunks_per_device; /* chunks of tile rows per device (0 = 4): finer chunks balance content whose cost varies over the image */
    uint32_t reserved;          /* 0 */
} llcomp_mi_opts;
#define LLCOMP_MI_MAX_DEVICES 64
#define LLCOMP_MI_FLAG_SMALL_MODEL 1u /* llcomp_mi_decode_flags / llcomp_mi_codec_create_ex */

/* ---- host-buffer API: drop-in for compressImage / decompressImage ------------------------------------ */
/* px: h*w*c bytes, row-major, channels interleaved (exactly the reference's `rgb` vector).  opts==NULL means
 * LEGACY on the current device.  *out is allocated by the library; release with llcomp_mi_free. */
int llcomp_mi_encode(const uint8_t* px, uint32_t w, uint32_t h, uint32_t c, const llcomp_mi_opts* opts,
                     uint8_t** out, size_t* out_len);
/* Accepts either wire format (dispatch on the magic byte).  *px allocated by the library. */
int llcomp_mi_decode(const uint8_t* data, size_t len, int32_t device, uint8_t** px, uint32_t* w, uint32_t* h,
                     uint32_t* c);
/* The same with flags: LLCOMP_MI_FLAG_SMALL_MODEL = a LEGACY stream that was written with the small model (a SLICED
 * container says so itself; the flag is ignored for it). */
int llcomp_mi_decode_flags(const uint8_t* data, size_t len, int32_t device, uint32_t flags, uint8_t** px, uint32_t* w,
                           uint32_t* h, uint32_t* c);
void llcomp_mi_free(void* p);
/* Decoding over a device list: the mirror image of llcomp_mi_opts.devices -- every device receives the table entries and payload
 * bytes of its chunks of tile rows, decodes them and copies its rows straight to their place in the picture.  Nothing is written to
 * the output before EVERY device has reported success (the first failing device in list order decides the status).  How the
 * container was encoded (one device or many, which chunking) does not matter.  A LEGACY stream goes to devices[0]; so does a
 * container whose slice table does not fit its payload (the one-device path forms the verdict for damaged input).
 * n_devices == 0 or devices == NULL: BAD_ARGS.  chunks_per_device: 0 = 4. */
int llcomp_mi_decode_devices(const uint8_t* data, size_t len, const int32_t* devices, uint32_t n_devices, uint32_t chunks_per_device,
                             uint32_t flags, uint8_t** px, uint32_t* w, uint32_t* h, uint32_t* c);
int llcomp_mi_decode_into_devices(const uint8_t* data, size_t len, const int32_t* devices, uint32_t n_devices, uint32_t chunks_per_device,
                                  uint32_t flags, uint8_t* px, size_t px_cap, uint32_t* w, uint32_t* h, uint32_t* c);
/* After a call of THIS thread returned LLCOMP_MI_DEVICE_FAILED: the HIP ordinal that failed, its position in the device list and the
 * status it reported (HIP_ERROR, NOMEM, NO_DEVICE, BAD_ARGS for an ordinal that does not exist).  Returns 0 and leaves the outputs
 * alone when the thread's last device-list call did not fail that way.  Any pointer may be NULL. */
int llcomp_mi_last_device_error(int32_t* device, uint32_t* index, int* status);
/* The same two calls with CALLER-PROVIDED output buffers (nothing is allocated for the caller).  If the capacity is too
 * small they return LLCOMP_MI_OUTPUT_OVERFLOW and report what it takes (*out_len; *w,*h,*c), and nothing is written.
 * Every host-buffer call works on a private HIP stream (never the NULL stream); with input and output buffers from
 * llcomp_mi_host_alloc (pinned memory) the PCIe copies are plain DMA, pageable buffers are staged by the HIP runtime. */
int llcomp_mi_encode_into(const uint8_t* px, uint32_t w, uint32_t h, uint32_t c, const llcomp_mi_opts* opts, uint8_t* out,
                          size_t out_cap, size_t* out_len);
int llcomp_mi_decode_into(const uint8_t* data, size_t len, int32_t device, uint8_t* px, size_t px_cap, uint32_t* w,
                          uint32_t* h, uint32_t* c);
/* ... with flags, as llcomp_mi_decode_flags (a LEGACY stream written with the small model) */
int llcomp_mi_decode_into_flags(const uint8_t* data, size_t len, int32_t device, uint32_t flags, uint8_t* px, size_t px_cap,
                                uint32_t* w, uint32_t* h, uint32_t* c);
/* The host-buffer calls keep a few idle coding lanes (GBs of HBM workspace for a 4K frame) for the next call of the same
 * shape, and the library parks the device memory of destroyed codecs / streams / lanes for reuse instead of returning it
 * to the driver (up to the pool limit per device; released by itself when an allocation OF THE LIBRARY fails).  This
 * releases both.  Codec and stream objects in use are not touched.  A process that shares the GPU with another allocator
 * (PyTorch's caching allocator, say) calls this when that allocator reports out-of-memory. */
void llcomp_mi_trim(void);
/* Parked device memory allowed PER DEVICE (default 16 GiB, or the environment's LLCOMP_MI_POOL_MAX_BYTES read once);
 * blocks beyond it go back to the driver, largest first.  0 = park nothing: every release is a hipFree. */
void llcomp_mi_set_pool_limit(uint64_t bytes_per_device);
uint64_t llcomp_mi_pool_limit(void);
uint64_t llcomp_mi_pool_idle_bytes(void); /* bytes parked right now, all devices */
void* llcomp_mi_host_alloc(size_t bytes); /* pinned host memory, NULL on failure */
void llcomp_mi_host_free(void* p);
const char* llcomp_mi_strerror(int status);
int llcomp_mi_abi_version(void);
/* Test / tuning hooks (LLCOMP_MI_LPW, _LANE_SHIFT, _NOROWS, _NOLDSTAB, _NOSNAP, _NOCACHE, _NOFEEDBACK, _OVERLAP, _FORCE_REPLAY; none
 * changes an output byte) are
 * read from the environment once per process; a test that changes them calls this to have them read again. */
void llcomp_mi_reload_tuning(void);
/* Number of usable HIP devices (0 when there is none; never fails). */
int llcomp_mi_device_count(void);

/* ---- host-side container tools (no GPU involved) ------------------------------------------------------ */
typedef struct llcomp_mi_info {
    uint32_t format, channels, width, height, tile_w, tile_h, planar, n_slices;
    uint64_t table_offset;   /* byte offset of the slice length table (0 for LEGACY) */
    uint64_t payload_offset; /* byte offset of the first payload byte */
    uint32_t small_model;    /* SLICED: the header's small-model flag; LEGACY: always 0 (not recorded in that header) */
    uint32_t reserved;
} llcomp_mi_info;
int llcomp_mi_probe(const uint8_t* data, size_t len, llcomp_mi_info* info);
uint32_t llcomp_mi_slice_count(uint32_t w, uint32_t h, uint32_t c, uint32_t tile_w, uint32_t tile_h, uint32_t planar);
/* Slice width for one-row slices (tile_h = 1) when `frames` frames are coded per call: the widest slice (64..480 pixels) that
 * still keeps about four wavefronts per SIMD busy.  A call that codes few frames is latency-bound with wide slices; this
 * trades a little compression (fresh models per slice) for it.  Returns 0 for nonsense arguments. */
uint32_t llcomp_mi_suggest_tile_w(uint32_t frames, uint32_t w, uint32_t h, uint32_t c, uint32_t planar);
/* FNV-1a-64 of a byte range (the checksum tests/golden records containers in); seed 0 starts a hash, a previous result
 * continues it over the next piece (header, slice table and payload of a container that lies in three buffers).  A running
 * hash that happens to be 0 (probability 2^-64 per piece) cannot be told from "start" and would restart: good enough for a
 * checksum of test vectors, not a keyed or adversarial hash. */
uint64_t llcomp_mi_fnv1a64(const uint8_t* data, size_t len, uint64_t seed);
/* Concatenator for multi-GPU sharding: `bands` are SLICED containers of consecutive horizontal bands of one
 * image (same width/channels/tile/planar; every band but the last a multiple of tile_h rows, because slices
 * have slice-local borders a band's slices ARE the full image's slices).  Produces the container of the whole
 * image.  Inverse: llcomp_mi_split_band extracts tile rows [tile_row0, tile_row1). */
int llcomp_mi_merge_bands(const uint8_t* const* bands, const size_t* band_lens, uint32_t n_bands, uint8_t** out,
                          size_t* out_len);
int llcomp_mi_split_band(const uint8_t* data, size_t len, uint32_t tile_row0, uint32_t tile_row1, uint8_t** out,
                         size_t* out_len);
/* The work split of every multi-GPU path (device lists here, ranks in llcomp_amd/sharding.py -- ONE implementation): the tile rows of
 * an image of `height` pixel rows in consecutive chunks, chunk i owned by part i % n_parts; chunks are as even as the tile grid allows,
 * and with fewer tile rows than n_parts * chunks_per_part every chunk is one tile row.  Writes (tile_row0, tile_row1, owner) triples
 * to `triples` (room for cap_chunks of them; NULL = only count) and the number of chunks to *n_chunks.  OUTPUT_OVERFLOW when
 * cap_chunks is too small (*n_chunks says what it takes).  tile_h 0 or > height = the whole height; chunks_per_part 0 = 4. */
int llcomp_mi_plan_chunks(uint32_t height, uint32_t tile_h, uint32_t n_parts, uint32_t chunks_per_part, uint32_t* triples,
                          uint32_t cap_chunks, uint32_t* n_chunks);

/* ---- device-resident batch codec: buffers stay in HBM, work is enqueued on the caller's stream --------- */
/* One codec object = fixed geometry (frames x h x w x c, tiling) + its own workspace on one device.
 * `frames` images of identical shape are coded per call; every frame gets the slices of the SLICED format
 * (slice ids run frame-major).  All device pointers are hipMalloc'ed (or torch) memory on that device.  */
typedef struct llcomp_mi_codec llcomp_mi_codec;
int llcomp_mi_codec_create(llcomp_mi_codec** codec, int32_t device, uint32_t frames, uint32_t w, uint32_t h,
                           uint32_t c, uint32_t tile_w, uint32_t tile_h, uint32_t planar);
/* flags: LLCOMP_MI_FLAG_SMALL_MODEL */
int llcomp_mi_codec_create_ex(llcomp_mi_codec** codec, int32_t device, uint32_t frames, uint32_t w, uint32_t h,
                              uint32_t c, uint32_t tile_w, uint32_t tile_h, uint32_t planar, uint32_t flags);
/* Does not wait for the device: the workspace is parked behind an event recorded on the stream of the codec's LAST encode /
 * decode (whether that call succeeded or not) and is handed out again only after it.  (One exception, microseconds in practice: a
 * codec destroyed right behind a 2-D decode waits for that call's 16-byte feedback copy, whose pinned mailbox it is about to free.)  That stream should outlive the work
 * queued on it; if it is destroyed earlier (legal HIP: hipStreamDestroy drains it in the background) the library notices the
 * dead event when the blocks are taken out again and drains the whole device instead. */
void llcomp_mi_codec_destroy(llcomp_mi_codec* codec);
uint32_t llcomp_mi_codec_slices(const llcomp_mi_codec* codec);        /* total = frames * slices per frame */
/* Diagnostic: which kernel family the codec's geometry selected when it was created -- bit 0 one-row slices (states on chip),
 * bit 1 one slice per wavefront (state table in LDS), bit 2 forced replay (test hook), bit 3 small model, bit 4 the encoder's state
 * snapshot pass, bit 5 the 2-D decoder's bank cache in LDS; bits 8..15 log2 of the lane-group width, bits 16..23 slices per
 * wavefront.  No effect on any output byte: tests use it to make sure they run the family they mean to. */
uint32_t llcomp_mi_codec_kernel_family(const llcomp_mi_codec* codec);
/* Device bytes the codec can hold at most.  The per-slice state tables (decoding 2-D slices; 63 KB per slice) and the snapshot
 * arrays of the 2-D encoder (22 B per sample) are allocated by the first call that needs them, so an encode-only or decode-only
 * codec stays below this figure; that first call can return LLCOMP_MI_NOMEM. */
uint64_t llcomp_mi_codec_workspace_bytes(const llcomp_mi_codec* codec);
/* Allocates NOW what the first encode (LLCOMP_MI_PREPARE_ENCODE: the 2-D encoder's snapshot arrays, or its state tables) and / or
 * the first decode (LLCOMP_MI_PREPARE_DECODE: the state tables of 2-D slices) would otherwise allocate inside the call -- for callers
 * that need the first call to be like every other one (no hipMalloc behind work already queued on their stream, no NOMEM in the
 * middle of a pipeline).  Idempotent; LLCOMP_MI_NOMEM when the device cannot give the memory. */
#define LLCOMP_MI_PREPARE_ENCODE 1u
#define LLCOMP_MI_PREPARE_DECODE 2u
int llcomp_mi_codec_prepare(llcomp_mi_codec* codec, uint32_t what);
/* Upper bound on the packed payload bytes the codec can emit for any input (13 B per sample + slack). */
uint64_t llcomp_mi_codec_max_payload_bytes(const llcomp_mi_codec* codec);
/* encode: d_px [frames][h][w][c] u8 -> d_payload (slice payloads packed back to back, slice order),
 * d_slice_len u32[slices], d_total u64[1] (= sum of lengths).  payload_cap = bytes available at d_payload; if
 * the packed size exceeds it nothing past the capacity is written and the status word reports OVERFLOW.
 * Asynchronous on `stream` (a hipStream_t, NULL = default stream).  d_status: u32[1], LLCOMP_MI_OK or an error,
 * valid once the stream has drained. */
int llcomp_mi_codec_encode(llcomp_mi_codec* codec, const void* d_px, void* d_payload, uint64_t payload_cap,
                           void* d_slice_len, void* d_total, void* d_status, void* stream);
/* decode: inverse.  d_payload/d_slice_len as produced by encode (payload_bytes = total), d_px out. */
int llcomp_mi_codec_decode(llcomp_mi_codec* codec, const void* d_payload, uint64_t payload_bytes,
                           const void* d_slice_len, void* d_px, void* d_status, void* stream);
/* Stage-A only (context + prediction model), for tests and profiling: d_sym u32[frames*h*w*c],
 * low 16 bits = folded context (0..7925), high 16 bits = folded residual (two's complement). */
int llcomp_mi_codec_model(llcomp_mi_codec* codec, const void* d_px, void* d_sym, void* stream);
/* Device-side concatenator (multi-GPU sharding: the gathering rank interleaves the ranks' packed payloads into image
 * order): copies n_seg byte ranges src[src_off[i] .. +len[i]) -> dst[dst_off[i] .. +len[i]) in one launch.  All five
 * pointers are device memory (offsets / lengths: u64[n_seg], computed on the GPU); any alignment; ranges must not overlap.
 * max_len = an upper bound of the lengths (sizes the grid only); n_seg <= 65535.  Asynchronous on `stream`. */
int llcomp_mi_device_copy_segments(const void* d_src, void* d_dst, const void* d_src_off, const void* d_dst_off,
                                   const void* d_len, uint32_t n_seg, uint64_t max_len, void* stream);
/* Sums over ranges of a u32 table in HBM: d_out[i] (u64) = sum of min(d_vals[j], cap) for j in [d_start[i], d_start[i] + d_count[i])
 * (d_start, d_count: u64[n] in HBM).  The multi-GPU path derives the byte counts of its (image, chunk) segments from the
 * slice-length tables with it.  Asynchronous on `stream`. */
int llcomp_mi_device_range_sums(const void* d_vals, const void* d_start, const void* d_count, void* d_out, uint32_t n, uint32_t cap,
                                void* stream);
/* The u32 status word written by encode/decode holds bit flags (1 overflow, 2 bad exponent, 4 truncated);
 * this maps it to an llcomp_mi_status. */
uint32_t llcomp_mi_status_from_bits(uint32_t bits);
/* Event counters of a codec object: what the rare and the adaptive paths of its kernels actually did, cumulative since creation (or
 * the last reset).  The kernels add to them with one atomic per wavefront -- and only wavefronts that have something to report -- so
 * they cost nothing measurable and are always on.  Tests use them to prove that a branch ran (parity passes either way); a caller can
 * watch the bank cache's hit rate on its content.  get_counters waits for the codec's last call (its own event, not the device),
 * writes the first n (<= LLCOMP_MI_CTR_COUNT) counters and clears all of them when reset != 0. */
enum {
    LLCOMP_MI_CTR_DEC_CACHED_WAVES = 0,    /* 2-D decoder (state tables in HBM): wavefronts that started with the bank cache in LDS */
    LLCOMP_MI_CTR_DEC_BYPASSED_WAVES = 1,  /* ... of those, the ones that gave it up (fewer than one hit in eight over four rows) */
    LLCOMP_MI_CTR_CACHE_LOOKUPS = 2,       /* state-bank look-ups in the cache (one per decoded sample while the cache is in use) */
    LLCOMP_MI_CTR_CACHE_MISSES = 3,        /* ... that missed: a 64-byte line fill from the table in HBM */
    LLCOMP_MI_CTR_CACHE_WRITEBACKS = 4,    /* victims written back to the table (one 32-byte sector each) */
    LLCOMP_MI_CTR_DEC_REPLAYS = 5,         /* decoded samples that ran out of window bytes (or saw an invalid exponent) on the fast path
                                              and went through rollback + checked replay (llcomp.hpp:219-247 is the checked form) */
    LLCOMP_MI_CTR_ENC_CARRY_BACKS = 6,     /* encoder: carries into a held 0xFF byte that went on into bytes already stored to HBM
                                              (the reference's outstanding_count run, llcomp.hpp:40-57, resolved eagerly) */
    LLCOMP_MI_CTR_GENERATION_WRAPS = 7,    /* state tables cleared because the 8-bit generation tag ran out (every 255 calls) */
    LLCOMP_MI_CTR_DEC_LAUNCHES_CACHED = 8, /* 2-D decode launches that ran with the bank cache */
    LLCOMP_MI_CTR_DEC_LAUNCHES_PLAIN = 9,  /* ... and without it, because (nearly) every wavefront of the last cached launch had given
                                              it up: the plain kernel holds no LDS for a cache nobody uses; re-probed every 16th call */
    LLCOMP_MI_CTR_COUNT = 16
};
int llcomp_mi_codec_get_counters(llcomp_mi_codec* codec, uint64_t* out, uint32_t n, int reset);
/* Per-kernel timing with hipEvents recorded on the caller's stream around each launch (bench.py's roofline leg).
 * get_profile drains the stream, adds up the milliseconds since the last call and resets:
 *   ms[0] state-table clear -- or, where the 2-D encoder replays its states ahead of the coder (slices of several rows and
 *         at most 4096 samples), the state snapshot pass that replaces the tables: k_snap_sort + _walk + _unperm
 *   ms[1] stage A (k_model_*)  ms[2] k_encode_slices  ms[3] k_scan_groups + k_pack_payload
 *   ms[4] k_group_sums + k_scan_groups + k_stage_streams (decode)  ms[5] k_decode_slices  ms[6] stage A inverse
 *   ms[7] state-table clear (decode) */
int llcomp_mi_codec_set_profiling(llcomp_mi_codec* codec, int enable);
int llcomp_mi_codec_get_profile(llcomp_mi_codec* codec, double* ms8, uint32_t* n_encode, uint32_t* n_decode);

/* ---- streaming pipeline: frames of one shape, host -> GPU -> host, several jobs in flight (BASELINE config 5) ------- */
/* The reference codes one image in RAM per call (llcompc.cpp:25-41, llcompd.cpp:17-31); this is the same operation as a
 * pipeline.  `depth` slots (1..16), each with its own codec object, HIP stream, HBM buffers and a pinned output buffer;
 * a job is one frame (SLICED container).  submit_* returns at once: LLCOMP_MI_OK, or LLCOMP_MI_BUSY when every slot is
 * occupied (back-pressure: take a result and release it).  `px` / `data` must stay valid until the job's result has
 * been returned by llcomp_mi_stream_wait; pinned memory (llcomp_mi_host_alloc, or the `data` of an earlier result that
 * has not been released) is copied by DMA while other jobs compute.  Results come back in submission order.  A
 * container that needs more than 2x the raw size fails with OUTPUT_OVERFLOW (llcomp_mi_encode handles such a frame).
 * One object is driven by one thread at a time (calls are serialised internally). */
typedef struct llcomp_mi_stream llcomp_mi_stream;
enum { LLCOMP_MI_JOB_ENCODE = 0, LLCOMP_MI_JOB_DECODE = 1 };
typedef struct llcomp_mi_stream_result {
    uint32_t slot;       /* hand back with llcomp_mi_stream_release when `data` is no longer needed */
    uint32_t kind;       /* LLCOMP_MI_JOB_ENCODE: data = container, LLCOMP_MI_JOB_DECODE: data = h*w*c pixels */
    int32_t status;      /* llcomp_mi_status of this job */
    uint32_t reserved;
    uint64_t tag;        /* the caller's tag from submit */
    const uint8_t* data; /* pinned host memory owned by the stream object; NULL when status != OK */
    uint64_t len;
} llcomp_mi_stream_result;
int llcomp_mi_stream_create(llcomp_mi_stream** stream, int32_t device, uint32_t w, uint32_t h, uint32_t c, uint32_t tile_w,
                            uint32_t tile_h, uint32_t planar, uint32_t depth);
/* Jobs of `frames_per_job` (1..64) frames: larger launches for the GPU, larger copies for the link.  submit_encode then
 * takes frames_per_job frames back to back, llcomp_mi_stream_submit_decode_batch that many containers; a result describes
 * the whole job (encode: all containers back to back, decode: all frames back to back) and llcomp_mi_stream_result_part
 * hands out container / frame f of a result that has been returned by wait and not yet released. */
int llcomp_mi_stream_create_ex(llcomp_mi_stream** stream, int32_t device, uint32_t w, uint32_t h, uint32_t c, uint32_t tile_w,
                               uint32_t tile_h, uint32_t planar, uint32_t depth, uint32_t frames_per_job);
/* The same pipeline over a device list (BASELINE config 5 "round-robin over the GPUs"): one pipeline of `depth` slots PER
 * DEVICE behind one object; jobs are dealt round-robin (a device whose slots are all occupied is skipped; BUSY when all are), results
 * still come back in submission order, every other call of this section works on the object unchanged (`slot` values are opaque).
 * An ordinal may repeat.  1 <= n_devices <= LLCOMP_MI_MAX_DEVICES.  A device that cannot be set up fails the call with
 * LLCOMP_MI_DEVICE_FAILED (llcomp_mi_last_device_error). */
int llcomp_mi_stream_create_multi(llcomp_mi_stream** stream, const int32_t* devices, uint32_t n_devices, uint32_t w, uint32_t h, uint32_t c,
                                  uint32_t tile_w, uint32_t tile_h, uint32_t planar, uint32_t depth, uint32_t frames_per_job);
uint32_t llcomp_mi_stream_devices(const llcomp_mi_stream* stream); /* pipelines behind the object (1 for a plain stream) */
uint32_t llcomp_mi_stream_frames_per_job(const llcomp_mi_stream* stream);
int llcomp_mi_stream_submit_decode_batch(llcomp_mi_stream* stream, const uint8_t* const* data, const size_t* lens, uint64_t tag);
int llcomp_mi_stream_result_part(llcomp_mi_stream* stream, uint32_t slot, uint32_t frame, const uint8_t** data, uint64_t* len);
void llcomp_mi_stream_destroy(llcomp_mi_stream* stream);
uint64_t llcomp_mi_stream_container_capacity(const llcomp_mi_stream* stream); /* largest container a slot can return */
int llcomp_mi_stream_submit_encode(llcomp_mi_stream* stream, const uint8_t* px, uint64_t tag);
int llcomp_mi_stream_submit_decode(llcomp_mi_stream* stream, const uint8_t* data, size_t len, uint64_t tag);
int llcomp_mi_stream_pending(llcomp_mi_stream* stream); /* jobs submitted and not yet returned by wait */
/* LLCOMP_MI_OK when llcomp_mi_stream_wait would not block (or nothing is pending), LLCOMP_MI_BUSY otherwise. */
int llcomp_mi_stream_poll(llcomp_mi_stream* stream);
/* Blocks until the OLDEST pending job has finished and describes it; BAD_ARGS when nothing is pending.  Single consumer:
 * wait / release / destroy of one pipeline object come from one thread (submits may come from another); the object's lock
 * is released while wait blocks. */
int llcomp_mi_stream_wait(llcomp_mi_stream* stream, llcomp_mi_stream_result* result);
int llcomp_mi_stream_release(llcomp_mi_stream* stream, uint32_t slot);

#ifdef __cplusplus
}
#endif
#endif

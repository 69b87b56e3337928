// kernels.hpp -- launchers of the gfx950 kernels (model_kernels.hip, slice_kernels.hip).  All launches are asynchronous on `stream`.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "geometry.hpp"

namespace llcomp_mi {

// status word bits written by kernels (atomicOr); mapped to llcomp_mi_status by the host
enum : uint32_t { kStOverflow = 1u, kStBadExponent = 2u, kStTruncated = 4u, kStInternal = 8u /* a kernel found its own assumptions violated */ };

// Event counters of a codec object (u64[kCtrCount] in HBM, cumulative; llcomp_mi_codec_get_counters, include/llcomp_mi.h
// LLCOMP_MI_CTR_*).  The kernels add to 0..6 with one atomic per wavefront and counter at their end -- and only wavefronts that have
// something to report; 7..9 are kept by the host (codec.hip).
enum : uint32_t {
    kCtrDecCachedWaves = 0,    // wavefronts of the 2-D decoder that started with the bank cache
    kCtrDecBypassedWaves = 1,  // ... of those, the ones that gave it up (fewer than one hit in eight)
    kCtrCacheLookups = 2,      // bank look-ups in the cache (lane-samples)
    kCtrCacheMisses = 3,       // ... that missed (line fill from the table in HBM)
    kCtrCacheWritebacks = 4,   // victims written back to the table
    kCtrDecReplays = 5,        // decoded samples that went through rollback + checked replay
    kCtrEncCarryBacks = 6,     // encoder carries that went on into bytes already stored to HBM
    kCtrGenerationWraps = 7,   // state tables cleared because the 8-bit generation ran out (host)
    kCtrDecLaunchesCached = 8, // 2-D decode launches with the bank cache (host)
    kCtrDecLaunchesPlainByFeedback = 9,  // ... and without it because the previous call's wavefronts all gave it up (host)
    kCtrCount = 16,
};

// Stage A (encode side): pixels u8 [frames][h][w][c] -> per-sample symbols u32, low 16 bits folded context, high
// 16 bits folded residual (llcomp.hpp:396-436).  Layout [frames][h][w][c] for interleaved slices, plane-major
// [frames][c][h][w] for planar slices (device_common.hpp), so that a slice row is contiguous.
hipError_t launch_model_fwd(const Geometry& g, const uint8_t* d_px, uint32_t* d_sym, hipStream_t stream);
// Stage A (decode side): reconstructed colour-transformed samples int16 -> pixels u8.  llcomp.hpp:532-543.
hipError_t launch_model_inv(const Geometry& g, const int16_t* d_rec, uint8_t* d_px, hipStream_t stream);

// Lane order: [lane_groups][slice_capacity_samples][64]; see model_kernels.hip.
uint32_t lane_groups(const Geometry& g);
uint32_t slice_capacity_samples(const Geometry& g);
hipError_t launch_to_lane_order_u32(const Geometry& g, const uint32_t* d_img, uint32_t* d_lanes, hipStream_t stream);
hipError_t launch_from_lane_order_i16(const Geometry& g, const int16_t* d_lanes, int16_t* d_img, hipStream_t stream);

// Planar 1-row slices: stage A fused with the lane-order transpose (pixels <-> lane-order arrays directly).
bool model_is_fused(const Geometry& g);
hipError_t launch_model_rows_fwd(const Geometry& g, const uint8_t* d_px, uint16_t* d_lanes, hipStream_t stream);  // 16-bit symbols
hipError_t launch_model_rows_inv(const Geometry& g, const int16_t* d_lanes, uint8_t* d_px, hipStream_t stream);

// 1-row slices keep their (three) contexts in LDS, and so does a launch with one slice per wavefront (the whole 63 KB
// table); only the other launches need the per-slice tables in HBM.
bool rows_mode(const Geometry& g);
bool slices_need_state_tables(const Geometry& g);

// One lane per slice: binarisation + adaptive states + range encoder.  llcomp.hpp:33-89, 166-206, 283-293, 439-449.
//   d_sym     : symbols in LANE ORDER: u32 (ctx | residual << 16), or the 16-bit form of the fused path when
//               model_is_fused(g)
//   d_states  : u64[lane group][kContexts][lanes of the group] (8 state bytes per context and slice), unused unless
//               slices_need_state_tables(g).  NOT cleared per call: every bank carries the `generation` (1..255) of the call
//               that wrote it in the spare top bits of its state bytes, and a bank of another generation reads as zeros.
//               The caller clears the table once (generation 0 = cleared memory) and again before it reuses a generation.
//   d_scratch : the slices' streams in stream lane order ; d_slice_len : u32[n_slices]
hipError_t launch_encode_slices(const Geometry& g, const void* d_sym, uint64_t* d_states, uint32_t generation, uint8_t* d_scratch,
                                uint32_t* d_slice_len, uint64_t* d_group_off, uint32_t* d_status, unsigned long long* d_counters,
                                hipStream_t stream);
// generation (1..255) -> the tag bits of a bank: bit i of the generation in the top bit of state byte i
uint64_t state_generation_tag(uint32_t generation);
// The snapshot coder of slices above 4096 samples, one segment (4096 samples of every slice) per launch: d_res / d_banks as
// launch_snapshot leaves them, d_seg_state = 64 bytes per slice in which a lane parks its coder between the segments.
hipError_t launch_encode_segment(const Geometry& g, const void* d_res, uint64_t* d_banks, uint8_t* d_scratch, uint32_t* d_slice_len,
                                 uint32_t* d_status, unsigned long long* d_counters, uint32_t seg_first, uint32_t* d_seg_state,
                                 hipStream_t stream);
// Offsets of the slices in the packed payload: one exclusive prefix value per LANE GROUP, u64[lane_groups + 1] (the last
// element and *d_total = sum of all lengths); pack / stage add the wave prefix of the group's own lengths.
// launch_encode_slices leaves the group sums in d_group_off itself when encoder_writes_group_sums(g); otherwise (and
// for decode, where the lengths come from outside) launch_group_sums computes them.  launch_scan_groups: sums -> offsets.
bool encoder_writes_group_sums(const Geometry& g);
hipError_t launch_group_sums(const Geometry& g, const uint32_t* d_slice_len, uint64_t* d_group_off, hipStream_t stream);
hipError_t launch_scan_groups(const Geometry& g, uint64_t* d_group_off, uint64_t* d_total, hipStream_t stream);
// u64[frames]: payload bytes of every frame (sum of its slices' lengths)
hipError_t launch_frame_bytes(const Geometry& g, const uint32_t* d_slice_len, uint64_t* d_frame_bytes, hipStream_t stream);
// Slice streams live in lane order for the serial kernels.  Behind the ENCODER: 16-byte units [group][unit][lane] (a lane flushes
// 16 bytes at a time); in front of the DECODER: dwords [group][dword][lane] (a lane reads a dword at a time: a running offset
// instead of unit arithmetic).  slice_cap/16 units
// per slice (model_kernels.hip).  pack: that order -> payload (slices back to back, capacity payload_cap);
// stage: payload -> that order.
hipError_t launch_pack_payload(const Geometry& g, const uint8_t* d_units, const uint32_t* d_slice_len,
                               const uint64_t* d_group_off, uint8_t* d_payload, uint64_t payload_cap,
                               uint32_t* d_status, hipStream_t stream);
hipError_t launch_stage_streams(const Geometry& g, const uint8_t* d_payload, uint64_t payload_bytes,
                                const uint32_t* d_slice_len, const uint64_t* d_group_off, uint8_t* d_units,
                                uint32_t* d_status, hipStream_t stream);
// n_seg byte ranges src[src_off[i] .. +len[i]) -> dst[dst_off[i] .. +len[i]) in one launch (offsets / lengths in HBM, any
// alignment); max_len = an upper bound of the lengths (grid sizing only).  The device-side concatenator of the multi-GPU path.
hipError_t launch_copy_segments(const uint8_t* d_src, uint8_t* d_dst, const uint64_t* d_src_off, const uint64_t* d_dst_off,
                                const uint64_t* d_len, uint32_t n_seg, uint64_t max_len, hipStream_t stream);
// out[i] = sum over j in [start[i], start[i] + count[i]) of min(vals[j], cap); everything in HBM.  (Multi-GPU path: byte counts
// of (image, chunk) segments from the slice-length tables.)
hipError_t launch_range_sums(const uint32_t* d_vals, const uint64_t* d_start, const uint64_t* d_count, uint64_t* d_out, uint32_t n,
                             uint32_t cap, hipStream_t stream);
// One lane per slice: range decoder + adaptive states + context model on reconstructed samples.
// llcomp.hpp:91-127, 219-247, 486-530.  d_rec int16 in LANE ORDER.
// d_units: the slices' streams in dword lane order (launch_stage_streams).  d_counters: kCtr* (may be null).  bank_cache = false:
// a geometry with kGeoBankCache runs the plain kernel this once (same bytes; the tables are per call, nothing carries over).
hipError_t launch_decode_slices(const Geometry& g, const uint8_t* d_units, const uint32_t* d_slice_len,
                                uint64_t* d_states, uint32_t generation, int16_t* d_rec, uint32_t* d_status, unsigned long long* d_counters,
                                bool bank_cache, hipStream_t stream);

}  // namespace llcomp_mi

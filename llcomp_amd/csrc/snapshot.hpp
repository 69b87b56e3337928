// snapshot.hpp -- the state snapshot pass of the 2-D encoder (snapshot_kernels.hip): launchers and workspace sizes.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "geometry.hpp"

namespace llcomp_mi {

// kGeoSnapshot is set (geometry.hpp) for slices of at most kSnapMaxSamples samples that share a wavefront and do not run the
// one-row kernels: slices of several rows, and one-row slices that miss those kernels (more than four channels, LLCOMP_MI_NOROWS).
bool snapshot_mode(const Geometry& g);
// ... and the slices are longer than one sorting capacity (4096 samples): the pass runs chunk after chunk, the states of a context
// are carried from chunk to chunk through the slice's state table in HBM (the decoder's: d_states with this call's generation)
bool snapshot_chunked(const Geometry& g);
// elements per array (all lane groups; sample capacity of one slice in the piece-layout arrays: snapshot_cap, geometry.hpp)
uint64_t snapshot_elems(const Geometry& g);
// d_sym: image-order symbols of stage A (launch_model_fwd).  Leaves, in piece layout [group][piece][lane][32 bytes]:
//   d_banks     u64 per sample: the eight states of the sample's context as they stand BEFORE the sample, stream order
//   d_residuals i16 per sample: the folded residual, stream order
// d_entries (u32 per sample) and d_sorted (u64 per sample) are scratch.
hipError_t launch_snapshot(const Geometry& g, const uint32_t* d_sym, void* d_entries, void* d_sorted, void* d_banks, void* d_residuals,
                           hipStream_t stream);
// snapshot_chunked(g): chunk c (samples c * 4096 ...) of every slice; call for c = 0, 1, ... in stream order.  d_ctx16 (u16 per sample:
// the context of every sorted position) and d_io (u64 per sample: the states a context run starts from) are scratch too, d_states /
// gpat = the codec's state tables and this call's generation tag (kernels.hpp).  The coder of chunk c (launch_encode_segment) depends
// on this chunk only.
hipError_t launch_snapshot_chunk(const Geometry& g, uint32_t c, const uint32_t* d_sym, void* d_entries, void* d_sorted, void* d_banks,
                                 void* d_residuals, void* d_ctx16, void* d_io, uint64_t* d_states, uint64_t gpat, hipStream_t stream);

}  // namespace llcomp_mi

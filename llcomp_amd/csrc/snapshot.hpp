// snapshot.hpp -- the state snapshot pass of the 2-D encoder (snapshot_kernels.hip): launchers and workspace sizes.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "geometry.hpp"

namespace llcomp_mi {

// kGeoSnapshot is set (geometry.hpp) for slices of at most kSnapMaxSamples samples that share a wavefront and do not run the
// one-row kernels: slices of several rows, and one-row slices that miss those kernels (more than four channels, LLCOMP_MI_NOROWS).
bool snapshot_mode(const Geometry& g);
// sample capacity of one slice in the piece-layout arrays (a multiple of 16) and elements per array (all lane groups)
uint32_t snapshot_cap(const Geometry& g);
uint64_t snapshot_elems(const Geometry& g);
// d_sym: image-order symbols of stage A (launch_model_fwd).  Leaves, in piece layout [group][piece][lane][32 bytes]:
//   d_banks     u64 per sample: the eight states of the sample's context as they stand BEFORE the sample, stream order
//   d_residuals i16 per sample: the folded residual, stream order
// d_entries (u32 per sample) and d_sorted (u64 per sample) are scratch.
hipError_t launch_snapshot(const Geometry& g, const uint32_t* d_sym, void* d_entries, void* d_sorted, void* d_banks, void* d_residuals,
                           hipStream_t stream);

}  // namespace llcomp_mi

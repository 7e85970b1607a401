// Kd-tree-free exact nearest-neighbour search on a two-level hashed voxel grid
// (replaces pcl::KdTreeFLANN behind getFitnessScore and GICP).
#pragma once
#include "common.hpp"

struct lsr_handle_s;

namespace lsr {
// N2: transform one keyframe (strided xyz, device) by a column-major 4x4 and write it at `offset` of `out`.
int transform_append(const void* d_aos, size_t stride_bytes, size_t n, const float* d_T16, DeviceCloud& out, size_t offset,
                     hipStream_t stream);
float nn_pick_cell(size_t n, const lsr_handle_s* h);
int nn_build_hash(const DeviceCloud& cloud, float cell, HashGridDev& grid, BuildScratch& sc, hipStream_t stream);
// the same structure for NDT targets whose voxel grid was built by the counting-sort builder: a refinement of the voxel order
// (fine cell = leaf / 8), one launch for up to LSR_GROUP targets, no host round trip
int nn_build_hash_from_grids(const VoxelGridDev* const* vgrids, HashGridDev* const* grids, int count, hipStream_t stream);
// mean squared 1-NN distance of T*source in the target, over pairs with d2 <= max_range.
int nn_fitness_score(const DeviceCloud& source, const float* T16_host, const HashGridDev& grid, double max_range, double* out,
                     BuildScratch& sc, DevBuf<float>& d_T16, hipStream_t stream);
// A/B switch for the wave-cooperative searches (env LSR_NN_COOP=0 selects the per-thread walks); read once.
bool nn_coop_enabled();
int nn_fitness_begin(const DeviceCloud& source, const float* T16_host, const HashGridDev& grid, double max_range, BuildScratch& sc,
                     DevBuf<float>& d_T16, hipStream_t stream);
int nn_fitness_end(BuildScratch& sc, hipStream_t stream, double* out);
// the search + reduction of a set of candidates in group launches on one stream (each member: its own scratch and mailbox)
struct FitJob { const DeviceCloud* source; const float* T16; const HashGridDev* grid; double max_range; BuildScratch* sc; };
int nn_fitness_begin_group(const FitJob* jobs, int count, hipStream_t stream);
// Device-side entry points (results stay in HBM): 1-NN of T*q (T nullable) and k-NN of q.
int nn_search_device(const DeviceCloud& q, const float* d_T16, const HashGridDev& grid, int fine_rings, float max_d2,
                     int* d_idx, float* d_d2, hipStream_t stream, int* d_work = nullptr);  // d_work: n + 1 ints => two-stage search
int knn_search_device(const DeviceCloud& q, const HashGridDev& grid, int k, int fine_rings, int* d_idx, float* d_d2,
                      hipStream_t stream);
int nn_search_host(const DeviceCloud& source, const float* T16_host, const HashGridDev& grid, int32_t* idx, float* d2,
                   BuildScratch& sc, DevBuf<float>& d_T16, hipStream_t stream);
}  // namespace lsr

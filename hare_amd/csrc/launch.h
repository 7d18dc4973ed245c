// launch.h -- declarations shared by api.cpp (the C-ABI), launch.cpp (kernel choice + launches) and device_scene.cpp (what a scene keeps
// on its device).  Product code; nothing from oracle/.
#pragma once
#include <string>

#include "../../include/hare_hip.h"
#include "scene.h"

namespace hare {

// a HIP call that must succeed: sets the thread-local message and returns HARE_E_NOMEM / HARE_E_HIP from the enclosing function
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            set_error(std::string(#expr) + " failed: " + (H->GetErrorString ? H->GetErrorString(_e) : "?")); \
            (void)H->GetLastError();                                                           \
            return (_e == hipErrorOutOfMemory) ? HARE_E_NOMEM : HARE_E_HIP;                    \
        }                                                                                      \
    } while (0)

constexpr unsigned kLdsMax = 160u * 1024u;
#ifndef HARE_K2P_WAVES_PER_EU
#define HARE_K2P_WAVES_PER_EU 4
#endif
#ifndef HARE_K2D_WAVES_PER_EU
#define HARE_K2D_WAVES_PER_EU 3      // K2d and the flags-only build of it (kernels.hip): 168 registers, nothing spilled
#endif

// ---- device_scene.cpp
int get_module(const HipApi* H, int device, const DeviceModule** out);
bool pool_can_serve(const Scene& s);                     // the pool kernel K1q can serve this grid (ct <= 512, bitmap + pools fit LDS)
int upload_cell_boxes(Scene& s, const HipApi* H);        // the voxels' tight boxes: only where they are used; never an error when they cannot be had
void upload_block_occ(Scene& s, const HipApi* H);        // option "voxel_skip": the block-level occupancy (device_scene.cpp)
void reserve_order_ring(Scene& s, const HipApi* H);      // the pool kernel's order ring (scene.h), sized by "voxel_order_max_rays"; called with every voxel build
void reserve_oct_scratch(Scene& s, const HipApi* H);     // the octree kernels' scratch ring, sized once for the largest launch the tree can get
int sync_partition_to_device(Scene& s, int kind);        // after a build: records, lists, tight boxes, kd device nodes

// ---- launch.cpp
bool ranges_overlap(const void* a, size_t na, const void* b, size_t nb);     // [a, a + na) and [b, b + nb) share a byte
void read_env_options(SceneOptions& o);
size_t voxel_scene_bytes(const Scene& s, size_t top);
enum class Kern { VoxelSimple, VoxelCount, VoxelAudit, VoxelProf, VoxelPool, VoxelPersist, VoxelOccl, OctSimple, OctCount, OctPool, OctPersist, OctDense, OctGroup, OctOccl,
                  KdSimple, KdCount, KdDense, None };
struct KernChoice {
    Kern k = Kern::None;
    const char* name = "";
    hipFunction_t f = nullptr;
};
// Which kernel serves a shoot: ONE function, used by the launcher and by hare_shoot_kernel_name (launch.cpp)
KernChoice choose_kernel(const Scene& s, const DeviceModule* M, int32_t kind, size_t top, int64_t n, uint32_t flags, bool flags_only = false);

}  // namespace hare

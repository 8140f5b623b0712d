// aclhip.hip -- gfx950 decode kernels and the host side of the C ABI of libaclhip.so (include/aclhip.h).
//
// One translation unit, in parts: the kernels live in kernels_*.inl, the host side in host_*.inl, included at the end of this file.
//
// Kernels (device helpers in aclhip_device.h):
//   decompress_tracks_kernel / decompress_tracks_any_settings_kernel
//       one wave64 per (clip instance, window of 312 pose quads = 104 tracks): scalar seek, base pose DMA'd global -> LDS, lanes <-> animated
//       sub-tracks decode in place into the LDS image (bit unpack, segment + clip range, W, lerp, normalize), the window streams
//       out 1 KiB per store instruction. The second entry point is the same body with per track rounding, the non default
//       default sub-track modes and always-normalize compiled in. Poses of several windows: decompress_tracks_in_turn_kernel (a wave
//       takes four work items in turn and keeps its LDS image while the clip stays the same; one 16 byte read of the bitstream per key).
//       Every launch is shaped by its batch (pose stride) and every kernel checks the clips it meets against that shape.
//   decompress_poses_consumer_kernel   the same decode, whole pose per wave, followed by what callers do next with a local pose -- blend K
//                                      clip instances, apply an additive clip onto its base, local -> object space -- before the pose leaves LDS.
//   decompress_track_kernel            one thread per (instance, bone) request; the registration time plan replaces the
//                                      reference's O(track index) skip over preceding widths.
//   decompress_scalar_tracks_kernel    scalar track lists: one wave64 per (instance, 256 tracks), lanes <-> tracks.
//   decompress_scalar_track_kernel     scalar track lists: one thread per (instance, track) request.
//   apply_tier_metadata_kernel         publishes / retires database tier metadata behind a stream ordered bulk copy.
//
// Every output store is a streaming store (store_streaming, aclhip_device.h): poses must not evict clip data from the L2s.
//
// Host side: context and clip / database registries (clips live in shared HBM slabs), blob validation, registration time tables,
// walk schedules of hierarchies, locality order of instance lists, launches.
//
// Build (acl_amd/build.py): hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -mllvm -amdgpu-kernarg-preload-count=16 aclhip.hip -ldl -o ../lib/libaclhip.so
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <shared_mutex>
#include <new>
#include <string>
#include <unordered_set>
#include <type_traits>
#include <vector>

#include "../../include/aclhip.h"
#include "acl_format.h"
#include "aclhip_device.h"

namespace aclhip
{
#include "kernels_pose.inl"
#if defined(ACLHIP_EXPERIMENTS)
#include "../../tools/experiments/kernels_experiments.inl"		// round 3's slower variants (profiles/r03_experiments.md); not part of a default build
#endif
#include "kernels_consumers.inl"
#include "kernels_misc.inl"
#include "kernels_scalar.inl"
#include "kernels_track.inl"
}

// ================================================================================================
// Host side: context, clip registry, C ABI
// ================================================================================================
using namespace aclhip;

#include "host_context.inl"
#include "host_clips.inl"
#include "host_databases.inl"
#if defined(ACLHIP_EXPERIMENTS)
#include "../../tools/experiments/host_experiments.inl"
#endif
#include "host_launch.inl"
#include "host_lists.inl"
#include "host_consumers.inl"
#include "host_scalar_misc.inl"

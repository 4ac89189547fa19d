/* vrc_internal.h -- launcher interface between vrc_api.hip (host logic) and
 * vrc_kernels.hip (gfx950 kernels).  Not part of the public C ABI. */
#ifndef VRC_INTERNAL_H
#define VRC_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vrc_core.h"

/* tf: 256 float4 (device).  lut: 256 float4 (device): (rgb*alpha', alpha'). */
hipError_t vrc_launch_build_lut( const float* tf, vrc_f4* lut, vrc_lut_params p,
                                 hipStream_t stream );

/* row-major brick (size voxels, elemBytes per voxel) -> micro-blocked atlas at slot origin */
hipError_t vrc_launch_repack_brick( const void* srcRowMajor, void* atlas, uint32_t elemBytes,
                                    const uint32_t size[3], const uint32_t slotVoxel[3],
                                    uint32_t nbx, uint32_t nby, hipStream_t stream );

/* inverse, for tests: atlas region -> row-major */
hipError_t vrc_launch_read_region( const void* atlas, void* dstRowMajor, uint32_t elemBytes,
                                   const uint32_t origin[3], const uint32_t size[3],
                                   uint32_t nbx, uint32_t nby, hipStream_t stream );

struct vrc_raycast_args
{
    vrc_frame frame;
    const vrc_dev_node* nodes;
    const int32_t* gridTable; /* NULL for the reference-order kernel */
    const void* atlas;
    const vrc_f4* lut;
    vrc_f4* pixelBuffer;
    unsigned long long* sampleCounter; /* NULL = do not count */
    bool clamp;
    bool gridDda;
};

hipError_t vrc_launch_raycast( const vrc_raycast_args& a, hipStream_t stream );

#endif

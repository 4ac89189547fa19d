/* vrc_internal.h -- launcher interface between vrc_api.hip (host logic) and
 * vrc_kernels.hip (gfx950 kernels).  Not part of the public C ABI. */
#ifndef VRC_INTERNAL_H
#define VRC_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vrc_core.h"

/* pixel tile of one wave64 */
#ifndef VRC_TILE_W
#define VRC_TILE_W 8u
#endif
#define VRC_TILE_H ( 64u / VRC_TILE_W )

/* Tile schedule.  The unit of the schedule is a 2x2 block of tiles (16x16 pixels): its four tiles take four
 * consecutive slots, and every raycast kernel runs four waves per workgroup, so the four waves of a
 * workgroup -- same CU, same vector L1 -- march neighbouring tiles and the micro-block lines two of them
 * share are fetched from L2 once (measured on C2 with the gather kernel: -3 % kernel time).  A frame whose
 * tile count is odd in x or y has units with fewer than four tiles: their spare slots hold VRC_NO_TILE.
 *
 * Round 4: the schedule is XCD-aware.  The dispatcher deals workgroups b, b + 1, ... round-robin over the 8 XCDs
 * (MI355X_MICROARCH.md, "Workgroup dispatch") and each XCD has its own L2: workgroup b takes unit (b / 8) of XCD
 * (b % 8)'s list.  A list is made of SUPER-TILES of VRC_SUPER_UNITS x VRC_SUPER_UNITS units (their units in Morton
 * order), dealt to the XCDs heaviest first in snake order.  Measured on C2 (profiles/r4_schedule_and_hbm_requests.txt):
 * the L2s fill whole 128-byte lines (TCC_EA0_RDREQ_128B = every request of these kernels) and fetch 1.9 x the packed (32-bit texel)
 * atlas per trilinear frame, 2.2 x the volume per point-sampled frame; super-tiles of 4 x 4 units (64 x 64 pixels on
 * one XCD) take 7 % of those requests away and no time (neighbouring workgroups drift apart in depth by more than the
 * few steps an L2 remembers), and cost the LDS-staged kernel 5 % (coarser heaviest-first order).  So the product keeps
 * super-tiles of ONE unit -- the heaviest-first order of rounds 1-3, with the XCD of every workgroup explicit -- and
 * the size stays a developer switch.  The schedule has VRC_XCDS x ceil(super-tiles / VRC_XCDS) x VRC_SUPER_UNITS^2
 * units; those outside the frame hold VRC_NO_TILE. */
#define VRC_NO_TILE 0xFFFFFFFFu
#define VRC_WAVES_PER_GROUP 4u
#ifndef VRC_SUPER_UNITS
#define VRC_SUPER_UNITS 1u
#endif
#define VRC_XCDS 8u
__host__ __device__ inline uint32_t vrc_super_x( uint32_t tilesX ) { return ( ( tilesX + 1u ) / 2u + VRC_SUPER_UNITS - 1u ) / VRC_SUPER_UNITS; }
__host__ __device__ inline uint32_t vrc_schedule_units( uint32_t tilesX, uint32_t tilesY )
{
    const uint32_t nSuper = vrc_super_x( tilesX ) * vrc_super_x( tilesY );
    return ( nSuper + VRC_XCDS - 1u ) / VRC_XCDS * VRC_XCDS * VRC_SUPER_UNITS * VRC_SUPER_UNITS;
}
__host__ __device__ inline uint32_t vrc_schedule_slots( uint32_t tilesX, uint32_t tilesY )
{
    return vrc_schedule_units( tilesX, tilesY ) * 4u;
}
/* tile of slot `slot` of unit `unit` (units row-major over the frame): VRC_NO_TILE outside the frame */
__host__ __device__ inline uint32_t vrc_unit_tile( uint32_t unit, uint32_t sub, uint32_t tilesX, uint32_t tilesY )
{
    const uint32_t unitsX = ( tilesX + 1u ) / 2u;
    const uint32_t tx = ( unit % unitsX ) * 2u + ( sub & 1u ), ty = ( unit / unitsX ) * 2u + ( sub >> 1 );
    return ( tx < tilesX && ty < tilesY ) ? ty * tilesX + tx : VRC_NO_TILE;
}
/* the tile a wave takes: from the schedule, or (no schedule) units in row-major order */
__device__ inline uint32_t vrc_slot_tile( const uint32_t* __restrict__ tileOrder, uint32_t slot, uint32_t tilesX,
                                          uint32_t tilesY )
{
    return tileOrder ? tileOrder[slot] : vrc_unit_tile( slot >> 2, slot & 3u, tilesX, tilesY );
}

/* tf: 256 float4 (device).  lut: VRC_TFP_ENTRIES float4 (device): the classified table
 * (rgb*alpha', alpha') with entries 256.. = 0, or with linear the padded transfer function. */
hipError_t vrc_launch_build_lut( const float* tf, vrc_f4* lut, vrc_lut_params p, bool linear,
                                 hipStream_t stream );

/* row-major brick (size voxels, elemBytes per voxel) -> micro-blocked slot (slot = device
 * pointer to the slot's first element; slotDim = padded slot size in voxels).  A brick smaller
 * than the slot gets its border voxels replicated into the padding. */
hipError_t vrc_launch_repack_brick( const void* srcRowMajor, void* slot, uint32_t elemBytes,
                                    const uint32_t size[3], const uint32_t slotDim[3],
                                    hipStream_t stream );

/* atlas -> tap-packed atlas (vrc_core.h): the packed texels of elements [firstElem, firstElem + nElems) of the atlas of
 * 8- or 16-bit voxels (whole slots; the packed atlas holds vrc_packed_elems( atlas elements ) texels of
 * VRC_PK_TEXEL( elemBytes ) bytes) */
hipError_t vrc_launch_pack_slots( const void* atlas, void* packed, uint64_t firstElem, uint64_t nElems,
                                  const uint32_t slotDim[3], uint32_t elemBytes, hipStream_t stream );

/* inverse, for tests: logical atlas region -> row-major */
hipError_t vrc_launch_read_region( const void* atlas, void* dstRowMajor, uint32_t elemBytes,
                                   const uint32_t origin[3], const uint32_t size[3],
                                   const vrc_layout& lay, hipStream_t stream );

/* histogram of the voxels [origin, origin+size) of one slot (slot-local coordinates): bins[v /
 * (typeRange / binCount)] += scale per voxel; bins is device memory, zeroed by the caller */
hipError_t vrc_launch_brick_histogram( const void* slot, uint32_t elemBytes, uint32_t sbx, uint32_t sby,
                                       const uint32_t origin[3], const uint32_t size[3],
                                       uint32_t binCount, unsigned long long scale,
                                       unsigned long long* bins, hipStream_t stream );

#define VRC_MAX_ERT_PARTS 8

struct vrc_raycast_args
{
    vrc_frame frame;
    const vrc_dev_node* nodes;
    const int32_t* gridTable; /* NULL for the reference-order kernel */
    const void* atlas;
    const vrc_f4* lut;
    vrc_f4* pixelBuffer;
    unsigned long long* sampleCounter; /* NULL = do not count */
    const uint32_t* tileOrder;         /* NULL = row-major tile order */
    bool clamp;
    bool gridDda;
    bool fixedStepping; /* VRC_OPT_STEPPING */
    bool linear;        /* VRC_OPT_FILTER = 1 (trilinear) */
    uint32_t elemBytes; /* voxel size: 1 (u8) or 2 (u16; classified per sample).  lut holds the padded
                         * transfer function whenever samples are classified one by one */
    vrc_classifier classifier;
    bool bigAtlas; /* more than 2^32 voxels: node slot bases are 64-bit (BIG kernel instances) */
    bool greyTable;    /* the transfer function is grey and the frame starts from zero: the two-float table form
                        * (VRC_MODE_GREY) of the point-sampling grid-walk kernel composites the same bits */
    int ertParts;      /* > 1: ray compaction, the march in this many launches (vrc_k_raycast_part); the host sets it
                        * only for the table-driven point-sampling walk kernel and frames below 65536 pixels a side */
    uint32_t* rayList; /* counts[VRC_MAX_ERT_PARTS] | two lists of width * height packed pixels */
    bool packed;     /* trilinear through the tap-packed atlas: atlas = the pool's packed atlas (vrc_march_segment_packed) */
    bool packedWide; /* ... of more than 4 GiB, or of an atlas of more than 2^32 voxels: 64-bit lane pointers instead of
                      * scalar base + 32-bit offset (BIG instances) */
    bool depthSplit; /* two waves per tile, near / far half of every ray (vrc_k_raycast_split): set by the host
                      * only when early ray termination cannot occur in this frame and the frame is cleared */
};

/* the frame's tile schedule (above): order: vrc_schedule_slots() uint32; scratch: VRC_TILE_SCRATCH_WORDS uint32;
 * bucket: one byte per super-tile (at most one per schedule slot) */
#define VRC_TILE_SCRATCH_WORDS 260u
hipError_t vrc_launch_tile_order( const vrc_frame& f, uint32_t* order, uint32_t* scratch,
                                  uint8_t* bucket, hipStream_t stream );

hipError_t vrc_launch_raycast( const vrc_raycast_args& a, hipStream_t stream );

/* LDS-staged form (vrc_kernels_lds.hip): needs gridTable, !clamp, 8x8 tiles */
hipError_t vrc_launch_raycast_lds( const vrc_raycast_args& a, hipStream_t stream );


/* per-ray adaptive LOD form (vrc_kernels_raylod.hip): gridTable = frame.lodLevels cell -> node
 * tables, lut = VRC_MAX_LOD_LEVELS classified tables of VRC_LUT_ENTRIES (u8 point sampling) or the
 * padded transfer function */
hipError_t vrc_launch_raycast_raylod( const vrc_raycast_args& a, hipStream_t stream );

/* shared by the translation units of libvrc_hip.so (vrc_api.hip owns the state) */
#include <string>
struct vrc_ctx;
int vrc_internal_fail( int code, const std::string& msg );          /* sets vrc_last_error, returns code */
void vrc_internal_note_kernel( const char* fmt, ... );               /* the kernel instance a launcher took (vrc_last_kernel) */
void vrc_internal_note_kernel_fn( const void* fn, int threads, size_t dynamicLds ); /* ... and its entry point (vrc_last_kernel_occupancy) */
hipStream_t vrc_internal_ctx_stream( vrc_ctx* ctx, int* deviceOut ); /* the context's render stream + device */

#endif

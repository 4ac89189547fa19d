/*
 * vrc_kernels_raylod.hip -- per-ray adaptive LOD form of the raycast (EXTENSION, BASELINE C5;
 * VRC_OPT_RAY_LOD).  The reference picks the LOD per brick on the host
 * (livre/core/render/SelectVisibles.cpp:52-68); here the node list is a hierarchy of resident
 * bricks and every ray applies the same screen-space-error criterion at the cells it crosses
 * (vrc_pixel_ray_lod in vrc_core.h has the definition).
 *
 * Same decomposition as vrc_k_raycast: one wave64 = one 8x8 pixel tile (Morton lanes), tiles
 * heaviest-first.  What differs:
 *   - the classified table exists once per level (opacity exponent alphaCorrection * 2^level),
 *     K * 257 entries in dynamic LDS, filled by the workgroup; four waves share one copy;
 *   - lanes of a wave may be in bricks of different levels and so step at different rates; the
 *     march itself is vrc_march_brick with the level's step and table.
 */
#include "vrc_internal.h"

#ifndef VRC_RL_WAVES
#define VRC_RL_WAVES 4u
#endif
#define VRC_RL_THREADS ( 64u * VRC_RL_WAVES )

#ifndef VRC_RL_MIN_BLOCKS
#define VRC_RL_MIN_BLOCKS 2
#endif
/* BIG: an atlas of more than 2^32 voxels (BASELINE C3's 22.6 GB): 64-bit slot bases, float positions, as the BIG
 * instances of vrc_k_raycast (round 4: per-ray LOD was not offered for such a pool) */
template < bool CLAMP, bool COUNT, bool FIXED, int MODE, typename ATLAS_T, bool BIG = false >
__global__ __launch_bounds__( VRC_RL_THREADS, VRC_RL_MIN_BLOCKS ) void vrc_k_raycast_raylod(
    const vrc_frame f, const vrc_dev_node* __restrict__ nodes,
    const int32_t* __restrict__ levelTables, const ATLAS_T* __restrict__ atlas,
    const vrc_f4* __restrict__ lutGlobal, const uint32_t lutEntries, const vrc_classifier cls,
    vrc_f4* __restrict__ pixelBuffer, unsigned long long* __restrict__ sampleCounter,
    const uint32_t* __restrict__ tileOrder, const uint32_t tilesX, const uint32_t nTiles )
{
    /* MODE_TABLE: lodLevels classified tables of 257 entries; else the padded transfer function */
    extern __shared__ __attribute__( ( aligned( 16 ) ) ) vrc_f4 lutLevels[]; /* 16: ds_read_b128 per entry */
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    if( MODE == VRC_MODE_PACKED || MODE == VRC_MODE_PACKED_GREY )
    {
        /* the tap-packed trilinear march (vrc_core.h): its classifier's table from the padded transfer function (the
         * level's opacity exponent travels in the classifier), and the packed atlas's per-axis byte offsets below */
        for( uint32_t i = tid; i < lutEntries; i += VRC_RL_THREADS )
            lutLevels[i] = vrc_cls8_entry( lutGlobal, i, MODE == VRC_MODE_PACKED_GREY );
    }
    else if( MODE == VRC_MODE_GREY )
    {
        /* grey transfer function: (grey, alpha) pairs, half the table bytes (vrc_core.h, VRC_MODE_GREY) */
        vrc_f2* const lut2 = reinterpret_cast< vrc_f2* >( lutLevels );
        for( uint32_t i = tid; i < lutEntries; i += VRC_RL_THREADS )
        {
            const vrc_f4 e = lutGlobal[i];
            lut2[i] = vrc_f2{ e.x, e.w };
        }
    }
    else
        for( uint32_t i = tid; i < lutEntries; i += VRC_RL_THREADS )
            lutLevels[i] = lutGlobal[i];
#if defined( VRC_ADDR_TABLES )
    if( MODE == VRC_MODE_PACKED || MODE == VRC_MODE_PACKED_GREY )
    {
        for( uint32_t u = tid; u < 256u; u += VRC_RL_THREADS )
        {
            constexpr uint32_t TB = sizeof( ATLAS_T ) == 8 ? 4u : 2u; /* bytes per texel (ATLAS_T: the packed mode's tag) */
            vrc_addr_tab[u] = TB * vrc_pk_x( u );
            vrc_addr_tab[256u + u] = TB * vrc_pk_y( u, f.sbx );
            vrc_addr_tab[512u + u] = TB * vrc_pk_z( u, f.sbx, f.sby );
        }
    }
    else if( FIXED )
    {
        const uint32_t cyy = f.sbx * VRC_MB_VOXELS - 64u, czz = f.sbx * f.sby * VRC_MB_VOXELS - 512u;
        for( uint32_t u = tid; u < 256u; u += VRC_RL_THREADS )
        {
            const uint32_t q = u >> VRC_MB_SHIFT;
            vrc_addr_tab[u] = u + 504u * q;
            vrc_addr_tab[256u + u] = 8u * u + cyy * q + VRC_MB_FIX_Y( u );
            vrc_addr_tab[512u + u] = 64u * u + czz * q - VRC_MB_FIX_Z( u );
        }
    }
#endif
    __syncthreads();
    const uint32_t slotIndex = blockIdx.x * VRC_RL_WAVES + ( tid >> 6 );
    const uint32_t tilesY = nTiles / tilesX;
    if( slotIndex >= vrc_schedule_slots( tilesX, tilesY ) )
        return;
    const uint32_t tile = vrc_slot_tile( tileOrder, slotIndex, tilesX, tilesY );
    if( tile == VRC_NO_TILE )
        return;
    const uint32_t tx = tile % tilesX, ty = tile / tilesX;
    const uint32_t lx = ( lane & 1u ) | ( ( lane >> 1 ) & 2u ) | ( ( lane >> 2 ) & 4u );
    const uint32_t ly = ( ( lane >> 1 ) & 1u ) | ( ( lane >> 2 ) & 2u ) | ( ( lane >> 3 ) & 4u );
    const uint32_t px = tx * 8u + lx;
    const uint32_t py = ty * 8u + ly;

    uint32_t nSamples = 0;
    if( px < f.width && py < f.height )
        vrc_pixel_ray_lod< CLAMP, COUNT, FIXED, MODE, ATLAS_T,
                           ( MODE == VRC_MODE_PACKED || MODE == VRC_MODE_PACKED_GREY )
                               ? ( sizeof( ATLAS_T ) == 8 ? VRC_PGROUP16 : VRC_PGROUP )
                               : VRC_GROUP,
                           BIG >(
            f, nodes, levelTables, atlas, lutLevels, cls, pixelBuffer, px, py, nSamples );
    if( COUNT )
    {
        unsigned long long s = nSamples;
#pragma unroll
        for( int off = 32; off > 0; off >>= 1 )
            s += __shfl_down( s, off, 64 );
        if( lane == 0 && s != 0 )
            atomicAdd( sampleCounter, s );
    }
}

template < bool CLAMP, bool COUNT, bool FIXED, int MODE, typename ATLAS_T, bool BIG = false >
static hipError_t launch_raylod( const vrc_raycast_args& a, hipStream_t stream )
{
    const uint32_t tilesX = ( a.frame.width + 7u ) / 8u, tilesY = ( a.frame.height + 7u ) / 8u;
    const uint32_t nTiles = tilesX * tilesY;
    if( nTiles == 0 )
        return hipSuccess;
    const uint32_t lutEntries = ( MODE == VRC_MODE_TABLE || MODE == VRC_MODE_GREY )
                                    ? a.frame.lodLevels * VRC_LUT_ENTRIES
                                    : ( ( MODE == VRC_MODE_PACKED || MODE == VRC_MODE_PACKED_GREY ) ? VRC_CLS8_ENTRIES : VRC_TFP_ENTRIES );
    vrc_internal_note_kernel( "vrc_k_raycast_raylod<%s,%s,%s,%d,%s,%s>", CLAMP ? "true" : "false", COUNT ? "true" : "false",
                              FIXED ? "true" : "false", (int)MODE,
                              sizeof( ATLAS_T ) == 1 ? "unsigned char" : ( sizeof( ATLAS_T ) == 2 ? "unsigned short" : ( sizeof( ATLAS_T ) == 4 ? "unsigned int" : "unsigned long" ) ),
                              BIG ? "true" : "false" );
    hipLaunchKernelGGL( ( vrc_k_raycast_raylod< CLAMP, COUNT, FIXED, MODE, ATLAS_T, BIG > ),
                        dim3( ( vrc_schedule_slots( tilesX, tilesY ) + VRC_RL_WAVES - 1u ) / VRC_RL_WAVES ),
                        dim3( VRC_RL_THREADS ),
                        lutEntries * ( MODE == VRC_MODE_GREY ? sizeof( vrc_f2 ) : sizeof( vrc_f4 ) ), stream, a.frame,
                        a.nodes, a.gridTable,
                        (const ATLAS_T*)a.atlas, a.lut, lutEntries, a.classifier, a.pixelBuffer,
                        a.sampleCounter, a.tileOrder, tilesX, nTiles );
    return hipGetLastError();
}

template < int MODE, typename ATLAS_T, bool BIG = false >
static hipError_t launch_raylod_classify( const vrc_raycast_args& a, bool count, hipStream_t stream )
{
    switch( ( a.clamp ? 2 : 0 ) | ( count ? 1 : 0 ) )
    {
    case 0: return launch_raylod< false, false, false, MODE, ATLAS_T, BIG >( a, stream );
    case 1: return launch_raylod< false, true, false, MODE, ATLAS_T, BIG >( a, stream );
    case 2: return launch_raylod< true, false, false, MODE, ATLAS_T, BIG >( a, stream );
    default: return launch_raylod< true, true, false, MODE, ATLAS_T, BIG >( a, stream );
    }
}

/* a.gridTable: lodLevels cell -> node tables; a.lut: lodLevels classified tables (u8 point
 * sampling) or the padded transfer function */
hipError_t vrc_launch_raycast_raylod( const vrc_raycast_args& a, hipStream_t stream )
{
    if( a.frame.lodLevels < 1 || a.frame.lodLevels > VRC_MAX_LOD_LEVELS || !a.gridTable ||
        a.frame.variant != VRC_VARIANT_CUDA )
        return hipErrorInvalidValue;
    const bool count = a.sampleCounter != nullptr;
    if( a.packed )
    {
        /* the trilinear filter through the tap-packed atlas (a.atlas), the hierarchy walk around it */
        if( !a.linear || ( a.elemBytes != 1 && a.elemBytes != 2 ) || a.clamp || ( a.bigAtlas && !a.packedWide ) )
            return hipErrorInvalidValue;
        /* (tags: uint32_t = the packed atlas of 8-bit voxels, uint64_t = of 16-bit voxels; ...,true>: 64-bit lane pointers) */
#define VRC_RL_PACKED( TAG, WIDE )                                                                                       \
    ( a.greyTable ? ( count ? launch_raylod< false, true, true, VRC_MODE_PACKED_GREY, TAG, WIDE >( a, stream )           \
                            : launch_raylod< false, false, true, VRC_MODE_PACKED_GREY, TAG, WIDE >( a, stream ) )        \
                  : ( count ? launch_raylod< false, true, true, VRC_MODE_PACKED, TAG, WIDE >( a, stream )                \
                            : launch_raylod< false, false, true, VRC_MODE_PACKED, TAG, WIDE >( a, stream ) ) )
        if( a.elemBytes == 2 )
            return a.packedWide ? VRC_RL_PACKED( uint64_t, true ) : VRC_RL_PACKED( uint64_t, false );
        return a.packedWide ? VRC_RL_PACKED( uint32_t, true ) : VRC_RL_PACKED( uint32_t, false );
#undef VRC_RL_PACKED
    }
    if( a.bigAtlas )
    {
        /* 64-bit slot bases: float positions (the classified tables of the levels for 8-bit point sampling) */
        if( a.elemBytes == 2 )
            return a.linear ? launch_raylod_classify< VRC_MODE_TRILINEAR, uint16_t, true >( a, count, stream )
                            : launch_raylod_classify< VRC_MODE_POINT, uint16_t, true >( a, count, stream );
        if( a.elemBytes != 1 )
            return hipErrorInvalidValue;
        return a.linear ? launch_raylod_classify< VRC_MODE_TRILINEAR, uint8_t, true >( a, count, stream )
                        : launch_raylod_classify< VRC_MODE_TABLE, uint8_t, true >( a, count, stream );
    }
    if( a.elemBytes == 2 )
        return a.linear ? launch_raylod_classify< VRC_MODE_TRILINEAR, uint16_t >( a, count, stream )
                        : launch_raylod_classify< VRC_MODE_POINT, uint16_t >( a, count, stream );
    if( a.elemBytes != 1 )
        return hipErrorInvalidValue;
    if( a.linear )
        return launch_raylod_classify< VRC_MODE_TRILINEAR, uint8_t >( a, count, stream );
    const bool fixed = a.fixedStepping && !a.clamp;
    switch( ( fixed ? 4 : 0 ) | ( a.clamp ? 2 : 0 ) | ( count ? 1 : 0 ) )
    {
    case 0: return launch_raylod< false, false, false, VRC_MODE_TABLE, uint8_t >( a, stream );
    case 1: return launch_raylod< false, true, false, VRC_MODE_TABLE, uint8_t >( a, stream );
    case 2: return launch_raylod< true, false, false, VRC_MODE_TABLE, uint8_t >( a, stream );
    case 3: return launch_raylod< true, true, false, VRC_MODE_TABLE, uint8_t >( a, stream );
    case 4:
        return a.greyTable ? launch_raylod< false, false, true, VRC_MODE_GREY, uint8_t >( a, stream )
                           : launch_raylod< false, false, true, VRC_MODE_TABLE, uint8_t >( a, stream );
    default:
        return a.greyTable ? launch_raylod< false, true, true, VRC_MODE_GREY, uint8_t >( a, stream )
                           : launch_raylod< false, true, true, VRC_MODE_TABLE, uint8_t >( a, stream );
    }
}

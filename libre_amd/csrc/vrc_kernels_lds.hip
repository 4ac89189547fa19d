/*
 * vrc_kernels_lds.hip -- LDS-staged form of the raycast kernel (gfx950).
 *
 * Same integrator and the same sample sequence as vrc_k_raycast (cuda/Renderer.cu:95-230),
 * but the voxels a wave needs for the next few steps are first copied into LDS with wide
 * coalesced loads and the march then reads LDS instead of issuing byte gathers:
 *
 *   - one wave64 = one 8x8 pixel tile (as before); a workgroup is VRC_LDS_WAVES independent
 *     waves that only share the transfer-function table;
 *   - a round = up to VRC_LDS_G steps of every lane that has a brick segment, served in up to
 *     VRC_LDS_PASSES passes.  A pass picks a lead lane (the tile centre if it still has steps to
 *     take), puts a region-sized window around the voxels the lead's next steps touch, and takes
 *     every lane of the lead's brick whose own steps stay inside that window.  ONE wave-wide OR
 *     (six DPP steps) of a packed bit mask -- slices | row pairs | 8-voxel pieces each lane
 *     touches, relative to the window -- gives the part of the window that is needed; that box is
 *     copied atlas -> LDS as 16-byte pieces (two 8-voxel rows of a micro-block slice, one
 *     global_load_dwordx4 + one ds_write2_b64 per lane and z-slice) and the samples are taken
 *     from LDS with a linear address (z*PZ + y*PY + x, compile-time pitches, so the eight
 *     trilinear taps are immediate offsets of one address);
 *   - lanes the first pass left out (the other brick of a tile that straddles a brick border,
 *     rays at another depth) get a pass of their own; only what is left after the last pass
 *     takes its steps by byte gathers.  Round 2 sent every such lane to the gathers: 8 gathers
 *     per trilinear sample for the whole wave whenever one lane was outside the box -- 29 % of
 *     the rounds on BASELINE C2, and the larger part of the texture addresser's time;
 *   - early ray termination is tested once per pass (opacity never decreases); a lane that
 *     crossed the threshold replays its steps of the pass one by one from the saved colour,
 *     which reproduces the reference's exit sample exactly (cuda/Renderer.cu:219-226);
 *   - the box of a pass has any shape of up to 264 rows of 32 voxels (rows per slice x slices);
 *   - a lead whose own steps do not fit the region halves the pass (8, 4, 2, 1 steps).
 *
 * Trilinear samples (VRC_OPT_FILTER = 1) are the reason this kernel exists: eight taps per
 * sample cost eight gathers in the gather form.  What a sample costs here was counted
 * instruction by instruction against profiles/r3_ubench_valu_lds_issue_costs.txt (wave64
 * fma/mul/add/and/lshr issue in ~1.15 ns per SIMD, conversions, min/max, 24-bit multiplies,
 * SDWA and DPP forms in ~1.85, log/exp in ~3.45; an LDS read at an odd address is 15 x slower
 * than an aligned one, so x0/x1 pair loads are out):
 *   - the fixed-point fractions are used as weights without the 2^-24 scale (W and 2^24 - W
 *     are exact floats; the scale of the three lerps, 2^-72, is folded into the classifier's
 *     multiplier -- same bits as scaled weights, three multiplies and nothing else less);
 *   - the transfer-function texel and its 1.8 fixed-point lerp weight come out of ONE float ->
 *     integer conversion (u = xB*256 + 256.5: texel = u >> 8, weight = u & 255), and for a grey
 *     transfer function both texels arrive in one ds_read_b128 of (g0, 1-a0, g1, 1-a1)/256.
 *
 * Requires overlap >= 1 (no clamped sampler), slot dims <= 248, a grid-aligned node set and
 * the CUDA lerp-weight width (VRC_OPT_TF_FRAC_BITS = 8) for the trilinear form;
 * vrc_api.hip / the launcher fall back to the gather kernels otherwise.
 *
 * Forms of the trilinear kernel (template arguments): 16-bit voxels (V: rows of 64 bytes, three
 * workgroups per CU), atlases of more than 2^32 voxels (BIG: 64-bit slot bases) and per-ray adaptive
 * LOD (RAYLOD: the grid walk replaced by the hierarchy hop of vrc_pixel_ray_lod, step and opacity
 * exponent per lane).  Point sampling through LDS exists for 8-bit voxels only (it reads the
 * classified table) and is a measured alternative, not what AUTO picks.
 *
 * The kernel is bound by vector issue (profiles/r3_rocprofv3_lds_trilinear_summary.txt: ~114 wave
 * instructions per 64-sample step, 60 of them the sample), so several constructs below are spelled
 * for the instruction they compile to: DPP reductions with old = 0, loads through pointers typed
 * address_space(1) behind an empty asm (scalar base + 32-bit lane offset), 24-bit multiplies,
 * position updates kept as additions.
 */
#include "vrc_internal.h"

#include <cstdlib>
#include <type_traits>

/* LDS region of one wave: VRC_LDS_REGION bytes = VRC_LDS_ROWS rows of VRC_LDS_PY voxels.  The row pitch is fixed
 * (an 8x8-pixel tile at up to 2 voxels per pixel spans 16 + 2 voxels, + 7 for the 8-voxel alignment of the pieces);
 * how the rows are split into rows per slice x slices is decided per pass from the box the lanes need: BASELINE C2's
 * tiles go from 10 x 10 voxels across at the front of the volume to 18 x 18 at its back (the pixel pitch doubles
 * with the distance from the eye) and march ~1 voxel per step along z, another view marches along y.  Round 2 had two
 * compile-time shapes (32x24x11, 32x20x16): on C2, 8 of a tile's 58 marching lanes did not fit the 11 slices of the
 * first in the average round (they are a voxel apart in depth) and took their steps by gathers; 16 slices in a
 * third shape with 16 rows lost as many at the back of the volume.  The price of the run-time slice pitch is one
 * address addition per trilinear sample (the second tap plane). */
#define VRC_LDS_PY 32u
#ifndef VRC_LDS_ROWS
#define VRC_LDS_ROWS 264u
#endif
#ifndef VRC_LDS_OCC
#define VRC_LDS_OCC 4 /* workgroups per CU the launch bounds ask for */
#endif
#ifndef VRC_LDS_STAGE_N
#define VRC_LDS_STAGE_N 11 /* slice loads in flight in one staging batch */
#endif
#define VRC_LDS_REGION ( VRC_LDS_PY * VRC_LDS_ROWS )
/* 16-bit voxels (trilinear only): rows of 32 voxels are 64 bytes, and the CU's 160 KiB of LDS decide how many waves
 * can hold a box at all.  Measured on the C2 shape with a 16-bit volume (1024^3, 128^3 bricks, 1024^2 pixels, ms per
 * frame, view along z / spun 30,20 deg; the gather form takes 2.93 / 3.80):
 *   264 rows, 2 workgroups per CU, 8 slices per staging batch   3.60 / 4.33
 *   264 rows, 2 per CU, 11 per batch                            3.07 / 4.00
 *   132 rows, 4 per CU,  5 per batch                            2.80 / 4.44      (7 per batch: 3.02 / 4.69)
 *   192 rows, 3 per CU,  8 per batch                            2.76 / 3.77
 *   192 rows, 3 per CU, 11 per batch                            2.38 / 3.74
 *   192 rows, 3 per CU, 10 per batch (no register spills)       2.37 / 3.53   <- taken   (9 per batch: 2.56 / 3.59)
 * (8-bit voxels: 264 rows at 4 per CU, 1.95 / 2.6.)  Twice the bytes per voxel cost the staged form most of its
 * advantage: either the boxes or the number of waves that hide each other's staging latency shrink. */
#ifndef VRC_LDS_ROWS16
#define VRC_LDS_ROWS16 192u
#endif
#ifndef VRC_LDS_OCC16
#define VRC_LDS_OCC16 3
#endif
#ifndef VRC_LDS_STAGE_N16
#define VRC_LDS_STAGE_N16 10 /* slice loads (of two 16-byte pieces) in flight in one staging batch */
#endif
#define VRC_LDS_MAX_DY 32u /* 16 row pairs: one per staging lane group */
#ifndef VRC_LDS_MAX_DZ
#define VRC_LDS_MAX_DZ ( 2u * VRC_LDS_STAGE_N ) /* two staging halves */
#endif
#ifndef VRC_LDS_KMAX
#define VRC_LDS_KMAX 8u /* steps of the lead a box is extended by at most (measured on C2, ms per frame: 4: 2.78, 6: 2.29,
                          * 7: 2.16, 8: 2.05, 12: 2.37, 15: 2.61, 20: 2.60 -- deep boxes are staged in two halves, one
                          * after the other, and the lanes of another brick take 8 steps per round whatever the box does) */
#endif
#define VRC_LDS_NMAX 30u /* steps a lane takes in one pass at most (one-hot in 32 bits) */
#ifndef VRC_LDS_WAVES
#define VRC_LDS_WAVES 4u
#endif
#ifndef VRC_LDS_G
#define VRC_LDS_G 8
#endif
#ifndef VRC_LDS_GF
#define VRC_LDS_GF VRC_LDS_G /* steps of one unrolled group of the march (developer builds: smaller groups with deeper boxes) */
#endif
#ifndef VRC_LDS_REFILL
#define VRC_LDS_REFILL 64 /* lanes without a next segment that trigger a walk by their number alone: 64 = never -- a walk
                           * runs when a lane is idle or about to need its next segment (measured on C2, ms per frame
                           * along z / off axis, mem and noise: 8: 1.82 / 2.48, 1.87 / 2.58; 16: 1.80 / 2.54, 1.82 / 2.53;
                           * 32: 1.77 / 2.42, 1.79 / 2.44; 64: 1.77 / 2.41, 1.78 / 2.43 -- the walk's ~220 instructions are
                           * the whole wave's, so the fewer the better) */
#endif
#ifndef VRC_LDS_SOON
#define VRC_LDS_SOON ( 2 * VRC_LDS_G ) /* a lane whose segment ends within this many steps needs its next one now */
#endif
#ifndef VRC_LDS_PASSES
#define VRC_LDS_PASSES 1 /* boxes per round before the remaining lanes fall back to gathers (measured on C2, round 3: 1: 2.38 ms, 2: 2.61, 3: 2.81) */
#endif

#if defined( VRC_LDS_STATS ) /* developer build only (tools/build_variants.sh) */
__device__ unsigned long long vrc_lds_stats[8];
extern "C" int vrc_debug_lds_stats( unsigned long long out[8], int reset )
{
    if( hipMemcpyFromSymbol( out, HIP_SYMBOL( vrc_lds_stats ), sizeof( unsigned long long ) * 8 ) != hipSuccess )
        return 1;
    if( reset )
    {
        const unsigned long long z[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        if( hipMemcpyToSymbol( HIP_SYMBOL( vrc_lds_stats ), z, sizeof( z ) ) != hipSuccess )
            return 1;
    }
    return 0;
}
#define VRC_LDS_STAT( I, V ) { if( lane == 0 ) atomicAdd( &vrc_lds_stats[I], (unsigned long long)( V ) ); }
#else
#define VRC_LDS_STAT( I, V ) {} /* a statement in both builds: `if( c ) VRC_LDS_STAT(..)` must not swallow what follows */
#endif
#if defined( VRC_LDS_TIMING ) /* developer build only */
/* wave cycles per phase (s_memtime; a wave's stalls are charged to the phase it stalls in) */
__device__ unsigned long long vrc_lds_phase[8];
extern "C" int vrc_debug_lds_phases( unsigned long long out[8], int reset )
{
    if( hipMemcpyFromSymbol( out, HIP_SYMBOL( vrc_lds_phase ), sizeof( unsigned long long ) * 8 ) != hipSuccess )
        return 1;
    if( reset )
    {
        const unsigned long long z[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        if( hipMemcpyToSymbol( HIP_SYMBOL( vrc_lds_phase ), z, sizeof( z ) ) != hipSuccess )
            return 1;
    }
    return 0;
}
#define VRC_LDS_PHASE( I ) { const unsigned long long now_ = __builtin_readcyclecounter(); phaseAcc[phaseCur] += now_ - phaseT0; phaseT0 = now_; phaseCur = I; }
#else
#define VRC_LDS_PHASE( I ) {}
#endif

namespace
{
/* wave64 OR: four DPP steps inside each row of 16 lanes, two row broadcasts, result read from lane 63 */
__device__ __forceinline__ uint32_t wave_or( uint32_t v )
{
    /* old = 0 (the identity of OR) with bound_ctrl: the compiler folds each step into ONE v_or_b32_dpp; with old = v
     * it emitted a copy, a v_mov_b32_dpp and the OR -- 18 instructions per reduction instead of 6, four reductions per
     * pass */
#define VRC_DPP_STEP( CTRL, ROWMASK ) \
    v |= (uint32_t)__builtin_amdgcn_update_dpp( 0, (int)v, CTRL, ROWMASK, 0xF, true );
    VRC_DPP_STEP( 0xB1, 0xF )  /* quad_perm [1,0,3,2] */
    VRC_DPP_STEP( 0x4E, 0xF )  /* quad_perm [2,3,0,1] */
    VRC_DPP_STEP( 0x141, 0xF ) /* row_half_mirror */
    VRC_DPP_STEP( 0x140, 0xF ) /* row_mirror */
    VRC_DPP_STEP( 0x142, 0xA ) /* row_bcast:15 into rows 1 and 3 */
    VRC_DPP_STEP( 0x143, 0xC ) /* row_bcast:31 into rows 2 and 3 */
#undef VRC_DPP_STEP
    return (uint32_t)__builtin_amdgcn_readlane( (int)v, 63 );
}

/* the staging loads: pointers into global memory, spelled out -- a slice's base goes through an empty asm (below), which
 * hides from the compiler where it points */
typedef uint32_t lds_u32x4 __attribute__( ( ext_vector_type( 4 ) ) );
typedef __attribute__( ( address_space( 1 ) ) ) const lds_u32x4 lds_g_u32x4;
typedef __attribute__( ( address_space( 1 ) ) ) const uint8_t lds_g_u8;
typedef __attribute__( ( address_space( 1 ) ) ) const uint16_t lds_g_u16;

/* copy N z-slices of the box: per lane one 16-byte piece (two 8-voxel rows) per slice */
/* between: work that does not depend on the box's voxels, done while the loads are on their way */
/* 16-bit voxels: the two rows are two 16-byte pieces (offsets, strides and pitches are in voxels) */
template < int N, typename F >
__device__ __forceinline__ void lds_stage( const uint16_t* __restrict__ slotPtr, uint32_t partial,
                                           uint32_t sliceStride, uint32_t z0, uint32_t dz, bool on,
                                           uint16_t* dst, uint32_t pz, F between )
{
    lds_u32x4 v0[N], v1[N];
    if( on )
    {
        uint32_t pl = partial;
#pragma unroll
        for( int z = 0; z < N; ++z )
        {
            const uint32_t zc = (uint32_t)z < dz ? (uint32_t)z : dz - 1u;
            const uint32_t zz = z0 + zc;
            const uint16_t* zb = slotPtr + ( ( zz >> VRC_MB_SHIFT ) * sliceStride + vrc_mb_z( zz ) );
            asm volatile( "" : "+s"( zb ), "+v"( pl ) ); /* see the 8-bit form */
            lds_g_u16* const g = (lds_g_u16*)zb;
            v0[z] = *(lds_g_u32x4*)( g + pl );
            v1[z] = *(lds_g_u32x4*)( g + pl + 8u );
        }
    }
    between();
    if( !on )
        return;
#pragma unroll
    for( int z = 0; z < N; ++z )
    {
        if( (uint32_t)z < dz )
        {
            /* (component by component: the struct copy kept the arrays in scratch memory) */
            *reinterpret_cast< uint4* >( dst + z * pz ) = make_uint4( v0[z].x, v0[z].y, v0[z].z, v0[z].w );
            *reinterpret_cast< uint4* >( dst + z * pz + VRC_LDS_PY ) = make_uint4( v1[z].x, v1[z].y, v1[z].z, v1[z].w );
        }
    }
}
template < int N, typename F >
__device__ __forceinline__ void lds_stage( const uint8_t* __restrict__ slotPtr, uint32_t partial,
                                           uint32_t sliceStride, uint32_t z0, uint32_t dz, bool on,
                                           uint8_t* dst, uint32_t pz, F between )
{
    lds_u32x4 v[N];
#if defined( VRC_LDS_LOAD_ALL ) /* developer build: lanes that stage nothing load too (the caller clamps their piece) */
    {
#else
    if( on )
    {
#endif
        uint32_t pl = partial;
#pragma unroll
        for( int z = 0; z < N; ++z )
        {
            const uint32_t zc = (uint32_t)z < dz ? (uint32_t)z : dz - 1u;
            const uint32_t zz = z0 + zc;
            const uint8_t* zb = slotPtr + ( ( zz >> VRC_MB_SHIFT ) * sliceStride + vrc_mb_z( zz ) );
            /* the slice's base is wave-uniform: kept in a scalar register pair (the empty asm stops the compiler from
             * re-associating it into slotPtr + partial + slice offset, a 64-bit vector addition per slice), so that the
             * load takes the scalar base + 32-bit lane offset form */
            /* (and the lane's offset goes through it too, one value from slice to slice: it is widened to 64 bits after
             * the asm, where the instruction selector can see that it is a 32-bit offset) */
            asm volatile( "" : "+s"( zb ), "+v"( pl ) );
            v[z] = *(lds_g_u32x4*)( (lds_g_u8*)zb + pl );
        }
    }
    between();
    if( !on )
        return;
#pragma unroll
    for( int z = 0; z < N; ++z )
    {
        if( (uint32_t)z < dz ) /* wave-uniform; never write past the box (the region ends with it) */
        {
            *reinterpret_cast< uint2* >( dst + z * pz ) = make_uint2( v[z].x, v[z].y );
            *reinterpret_cast< uint2* >( dst + z * pz + VRC_LDS_PY ) = make_uint2( v[z].z, v[z].w );
        }
    }
}

struct lds_box
{
    uint32_t x0, y0, z0; /* origin: x0 multiple of 8, y0 even */
    uint32_t dx, dy, dz; /* extents */
};

/* Transfer function and opacity correction of one trilinear sample (the oracle's orc_tf_fetch +
 * composite, cuda/ColorMap.cu:40-45, cuda/Renderer.cu:83-93; vrc_classify in vrc_core.h is the same
 * function in float form).  tq = xB*256 + 256.5 with xB = u*256 - 0.5 the CUDA texel coordinate
 * (already clamped to [0.5, 65792.25]): texel pair (tq >> 8, +1) of the padded table, lerp weight
 * (tq & 255)/256 = CUDA's 1.8 fixed-point weight, rounded to nearest as there.  tab holds, per
 * texel j, (colour_j, 1 - alpha_j) / 256: the weights are used as integers 0..256.
 * GREY: (g_j, 1-a_j, g_j+1, 1-a_j+1)/256 in one 16-byte entry. */
struct lds_cls
{
    float mult, add, kexp;
};
__device__ __forceinline__ float lds_alpha( float corr, float kexp )
{
    /* 1 - min(a, 255/256) = max(1 - a, 1/256) (Renderer.cu:88); pow as exp2(k log2 x) */
    corr = fmaxf( corr, 1.0f / 256.0f );
    return 1.0f - __builtin_amdgcn_exp2f( kexp * __builtin_amdgcn_logf( corr ) );
}
__device__ __forceinline__ vrc_f2 lds_classify( const vrc_f2*, const float4* tab, float d, const lds_cls& k )
{
    float tq = __builtin_fmaf( d, k.mult, k.add );
    tq = __builtin_amdgcn_fmed3f( tq, 0.5f, 65792.25f );
    const uint32_t u = (uint32_t)tq;
    const float a = (float)( u & 255u ), b = 256.0f - a;
    const float4 t = *reinterpret_cast< const float4* >( reinterpret_cast< const char* >( tab ) +
                                                        ( ( u >> 4 ) & 0xFFFF0u ) );
    const float alpha = lds_alpha( __builtin_fmaf( a, t.w, b * t.y ), k.kexp );
    vrc_f2 e;
    e.x = __builtin_fmaf( a, t.z, b * t.x ) * alpha;
    e.w = alpha;
    return e;
}
__device__ __forceinline__ vrc_f4 lds_classify( const vrc_f4*, const float4* tab, float d, const lds_cls& k )
{
    float tq = __builtin_fmaf( d, k.mult, k.add );
    tq = __builtin_amdgcn_fmed3f( tq, 0.5f, 65792.25f );
    const uint32_t u = (uint32_t)tq;
    const float a = (float)( u & 255u ), b = 256.0f - a;
    const float4* const p = reinterpret_cast< const float4* >( reinterpret_cast< const char* >( tab ) +
                                                               ( ( u >> 4 ) & 0xFFFF0u ) );
    const float4 t0 = p[0], t1 = p[1];
    const float alpha = lds_alpha( __builtin_fmaf( a, t1.w, b * t0.w ), k.kexp );
    vrc_f4 e;
    e.x = __builtin_fmaf( a, t1.x, b * t0.x ) * alpha;
    e.y = __builtin_fmaf( a, t1.y, b * t0.y ) * alpha;
    e.z = __builtin_fmaf( a, t1.z, b * t0.z ) * alpha;
    e.w = alpha;
    return e;
}

/* the eight taps at p, weights = the 24 fraction bits of the sample's coordinates, NOT scaled by 2^-24:
 * W and 2^24 - W are exact, so every product and sum is 2^24 (2^48, 2^72) times the one with scaled
 * weights, bit for bit; lds_cls.mult carries the 2^-72 */
/* p: the sample's voxel in the box, q: the same voxel one slice further */
template < typename V >
__device__ __forceinline__ void lds_taps( const V* p, const V* q, float t[8] )
{
    t[0] = (float)p[0];
    t[1] = (float)p[1];
    t[2] = (float)p[VRC_LDS_PY];
    t[3] = (float)p[VRC_LDS_PY + 1u];
    t[4] = (float)q[0];
    t[5] = (float)q[1];
    t[6] = (float)q[VRC_LDS_PY];
    t[7] = (float)q[VRC_LDS_PY + 1u];
}
__device__ __forceinline__ float lds_trilerp( const float v[8], uint32_t fx, uint32_t fy, uint32_t fz )
{
    const float wx = (float)( fx & 0xFFFFFFu ), wy = (float)( fy & 0xFFFFFFu ), wz = (float)( fz & 0xFFFFFFu );
    const float ux = 16777216.0f - wx, uy = 16777216.0f - wy, uz = 16777216.0f - wz;
    /* the oracle's order: x, then y, then z; a*(1-w) + b*w */
    const float c00 = __builtin_fmaf( v[1], wx, v[0] * ux );
    const float c10 = __builtin_fmaf( v[3], wx, v[2] * ux );
    const float c01 = __builtin_fmaf( v[5], wx, v[4] * ux );
    const float c11 = __builtin_fmaf( v[7], wx, v[6] * ux );
    const float c0 = __builtin_fmaf( c10, wy, c00 * uy );
    const float c1 = __builtin_fmaf( c11, wy, c01 * uy );
    return __builtin_fmaf( c1, wz, c0 * uz );
}
}

/* GREY: the transfer function is grey and the frame starts from zero (vrc_raycast_args.greyTable): colours and table
 * entries are (grey, alpha) pairs, bit-identical to the four-float form (vrc_core.h, VRC_MODE_GREY) */
/* RAYLOD: per-ray adaptive LOD (vrc_pixel_ray_lod in vrc_core.h; trilinear only: samples are classified one by one, so
 * no per-level table is needed): gridTable holds the per-level cell -> node tables, the walk is the hop of
 * vrc_ray_lod_hop, and a lane's step and opacity exponent are those of its brick's level */
/* V: the voxel type of the atlas (uint16_t: trilinear only -- a 16-bit density does not index the classified table) */
/* BIG: an atlas of more than 2^32 voxels -- a brick's slot base has 64 bits (trilinear only, like the 16-bit form) */
template < bool COUNT, bool LINEAR, bool GREY = false, bool RAYLOD = false, typename V = uint8_t, bool BIG = false >
/* four workgroups per CU: 4 x (4 regions of 8.25 KiB + the table) = 152 of the CU's 160 KiB */
__global__ __launch_bounds__( 64 * VRC_LDS_WAVES, sizeof( V ) == 2 ? VRC_LDS_OCC16 : VRC_LDS_OCC ) void vrc_k_raycast_lds(
    const vrc_frame f, const vrc_dev_node* __restrict__ nodes,
    const int32_t* __restrict__ gridTable, const V* __restrict__ atlas,
    const vrc_f4* __restrict__ lutGlobal, const vrc_classifier cls,
    vrc_f4* __restrict__ pixelBuffer, unsigned long long* __restrict__ sampleCounter,
    const uint32_t* __restrict__ tileOrder, const uint32_t tilesX, const uint32_t nTiles )
{
    using C = std::conditional_t< GREY, vrc_f2, vrc_f4 >; /* a colour / a table entry */
    /* point sampling: the classified table (257 entries of C).  Trilinear: the padded transfer function
     * tfp[j] = tf[clamp(j-1)] as (colour, 1 - alpha) / 256 per texel (lds_classify); grey: texels j and j+1
     * side by side in one 16-byte entry.  One texel more than the 258 of the float form: a sample at the
     * upper clamp reads texel pair (257, 258) with weights (256, 0). */
    constexpr uint32_t TAB_ENTRIES = LINEAR ? ( GREY ? VRC_TFP_ENTRIES : VRC_TFP_ENTRIES + 1u ) : 1u;
    __shared__ C lut[LINEAR ? 1u : VRC_TFP_ENTRIES];
    __shared__ __attribute__( ( aligned( 16 ) ) ) float4 tab[TAB_ENTRIES];
    static_assert( sizeof( V ) == 1 || LINEAR, "16-bit voxels are classified sample by sample" );
    static_assert( !BIG || ( LINEAR && !RAYLOD ), "64-bit slot bases: the trilinear form only" );
    constexpr uint32_t ROWS = sizeof( V ) == 2 ? VRC_LDS_ROWS16 : VRC_LDS_ROWS;
    __shared__ __attribute__( ( aligned( 16 ) ) ) V regions[VRC_LDS_WAVES][VRC_LDS_PY * ROWS];

    if constexpr( LINEAR )
    {
        for( uint32_t i = threadIdx.x; i < TAB_ENTRIES; i += 64u * VRC_LDS_WAVES )
        {
            const float sc = 1.0f / 256.0f;
            const vrc_f4 e0 = lutGlobal[i < VRC_TFP_ENTRIES - 1u ? i : VRC_TFP_ENTRIES - 1u];
            const vrc_f4 e1 = lutGlobal[i + 1u < VRC_TFP_ENTRIES - 1u ? i + 1u : VRC_TFP_ENTRIES - 1u];
            if constexpr( GREY )
                tab[i] = float4{ e0.x * sc, ( 1.0f - e0.w ) * sc, e1.x * sc, ( 1.0f - e1.w ) * sc };
            else
                tab[i] = float4{ e0.x * sc, e0.y * sc, e0.z * sc, ( 1.0f - e0.w ) * sc };
        }
    }
    else
    {
        for( uint32_t i = threadIdx.x; i < VRC_TFP_ENTRIES; i += 64u * VRC_LDS_WAVES )
        {
            const vrc_f4 e = lutGlobal[i];
            if constexpr( GREY )
                lut[i] = vrc_f2{ e.x, e.w };
            else
                lut[i] = e;
        }
    }
    __syncthreads();
    /* from here on the waves of the workgroup are independent: no further barrier */

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t slot = blockIdx.x * VRC_LDS_WAVES + wave;
    const uint32_t tilesY = nTiles / tilesX;
    if( slot >= vrc_schedule_slots( tilesX, tilesY ) )
        return;
    V* const region = regions[wave];
#if defined( VRC_LDS_TIMING )
    unsigned long long phaseAcc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, phaseT0 = __builtin_readcyclecounter();
    int phaseCur = 6;
#endif

    const uint32_t tile = vrc_slot_tile( tileOrder, slot, tilesX, tilesY );
    if( tile == VRC_NO_TILE ) /* the waves of a workgroup are independent from here on */
        return;
    const uint32_t tx = tile % tilesX, ty = tile / tilesX;
    const uint32_t lx = ( lane & 1u ) | ( ( lane >> 1 ) & 2u ) | ( ( lane >> 2 ) & 4u );
    const uint32_t ly = ( ( lane >> 1 ) & 1u ) | ( ( lane >> 2 ) & 2u ) | ( ( lane >> 3 ) & 4u );
    const uint32_t px = tx * 8u + lx, py = ty * 8u + ly;
    const bool inFrame = px < f.width && py < f.height;
    const uint32_t pixelPos = py * f.width + px;

    /* ---- ray set-up: vrc_pixel_grid_dda ------------------------------------------------ */
    vrc_ray r;
    C color = {};
    bool done = true; /* nothing (more) to do for this lane */
    bool store = false;
    if( inFrame )
    {
        r = vrc_setup_ray( f, px, f.rowMap ? f.rowMap[py] : py );
        if( r.hit )
        {
            if constexpr( !GREY ) /* the grey form is only taken for frames that start from zero */
                if( !f.clearFirst )
                    color = pixelBuffer[pixelPos];
            if( !( color.w > VRC_EARLY_EXIT ) )
            {
                store = true;
                done = false;
            }
        }
        else if( f.clearFirst )
            store = true; /* the folded clear: a missed pixel is written as 0 */
    }

    int cell[3] = { 0, 0, 0 };
    float tMax[3] = { 0.f, 0.f, 0.f };
    float t0 = 0.0f, t1 = 0.0f;
    uint32_t back = 0u;
    bool ddaEnd = false;
    [[maybe_unused]] float te = 0.0f; /* RAYLOD: where the next hop starts */
    if constexpr( RAYLOD )
    {
        if( !done )
        {
            if( vrc_ray_lod_interval( f, r, t0, t1 ) )
                te = t0;
            else
                done = true;
        }
    }
    else if( !done )
    {
        const vrc_f3 gmin = { f.gridMin[0], f.gridMin[1], f.gridMin[2] };
        const vrc_f3 gmax = { f.gridMin[0] + f.cellSize[0] * (float)f.gridDim[0],
                              f.gridMin[1] + f.cellSize[1] * (float)f.gridDim[1],
                              f.gridMin[2] + f.cellSize[2] * (float)f.gridDim[2] };
        const bool any = vrc_intersect_box( r.origin, r.invDir, gmin, gmax, &t0, &t1 );
        /* the GLSL twin does not clamp a brick's interval to the global box (fragRaycast.glsl:149-150):
         * bricks of the tree that reach past the volume (ragged trees) are sampled there too */
        if( f.variant == VRC_VARIANT_GL )
            t0 = fmaxf( t0, fmaxf( r.tNearPlane, 0.0f ) );
        else
        {
            t0 = fmaxf( fmaxf( t0, r.tNearGlobal ), fmaxf( r.tNearPlane, 0.0f ) );
            t1 = fminf( t1, r.tFarGlobal );
        }
        if( !( any && t0 <= t1 ) )
            done = true;
        else
        {
            const float o[3] = { r.origin.x, r.origin.y, r.origin.z };
            const float d[3] = { r.dir.x, r.dir.y, r.dir.z };
            const float id[3] = { r.invDir.x, r.invDir.y, r.invDir.z };
            /* as vrc_pixel_grid_dda (vrc_core.h): on a cell face at t0 the walk starts in the cell the ray
             * goes on into; `back` = the axes of those faces (the cells behind them are touched too) */
            const float tolE = fabsf( t0 ) * 2e-6f;
#pragma unroll
            for( int a = 0; a < 3; ++a )
            {
                const float p = o[a] + d[a] * t0;
                const float u = ( p - f.gridMin[a] ) * f.invCellSize[a];
                const bool pos = d[a] > 0.0f;
                const float kf = rintf( u );
                const float tFace = ( ( f.gridMin[a] + f.cellSize[a] * kf ) - o[a] ) * id[a];
                int c = (int)floorf( u );
                const int cOn = (int)kf - ( pos ? 0 : 1 );
                if( fabsf( tFace - t0 ) <= tolE && cOn >= 0 && cOn <= f.gridDim[a] - 1 )
                {
                    c = cOn;
                    back |= 1u << a;
                }
                c = c < 0 ? 0 : ( c > f.gridDim[a] - 1 ? f.gridDim[a] - 1 : c );
                cell[a] = c;
                const float boundary = f.gridMin[a] + f.cellSize[a] * (float)( pos ? c + 1 : c );
                tMax[a] = ( boundary - o[a] ) * id[a];
            }
        }
    }

    /* ---- per-lane segment state ---------------------------------------------------------- */
    /* A lane marches its CURRENT brick segment and holds the NEXT one ready (round 3).  Round 2 walked to the next
     * brick only when a lane had run out of steps, and batched those walks (the walk is ~500 instructions for the whole
     * wave): a lane that left its brick through a side face -- on BASELINE C2 the brick borders sweep across a third of
     * the rays inside every slab of bricks -- waited for company while the rest of its tile marched on, stayed one or
     * more rounds behind them in depth for the rest of the slab, fell outside their box round after round and took its
     * steps by gathers (30 % of the rounds had such lanes; the gather path was 0.68 of the kernel's 2.36 ms).  With
     * the next segment in registers a lane never waits, the walks stay batched (they now run AHEAD of need), and the
     * lanes of a tile stay within a step of each other in depth. */
    bool hasSeg = false, hasPend = false, walkDone = done;
    int32_t curNode = -1, pNode = -1;
    uint32_t pfx = 0, pfy = 0, pfz = 0, pfdx = 0, pfdy = 0, pfdz = 0, pSlotBase = 0;
    [[maybe_unused]] uint32_t pSlotHi = 0, laneSlotHi = 0; /* BIG: the high halves of the slot bases */
    float pTravel = 0.0f;
    /* RAYLOD: 2^level of the current and of the next segment's brick (step and opacity exponent scale with it) */
    [[maybe_unused]] float lscale = 1.0f, pScale = 1.0f;
    /* bricks already handed to the slab test; probe: what the walk does next at its cell (0: the cell
     * itself, 1..6: the cells around an edge / corner the ray leaves through, see vrc_pixel_grid_dda) */
    int32_t recent0 = -1, recent1 = -1, recent2 = -1, recent3 = -1;
    /* work list of cells around an edge / corner (encoding: vrc_pixel_grid_dda), the faces the ray leaves
     * the current cell through, and whether the step through them is still to be taken */
    /* bits 0-13: the work list; bits 16-18: tied axes; bit 20: the step through them is still to be taken
     * (one register: the trilinear form is short of them) */
    uint32_t walk = (uint32_t)( ( 0x7F68544032201000ull >> ( back * 8u ) ) & 0x7Fu );
    uint32_t fx = 0, fy = 0, fz = 0, fdx = 0, fdy = 0, fdz = 0; /* 8.24 slot-local voxel */
    float rdx = 0.0f, rdy = 0.0f, rdz = 0.0f;                   /* 1 / |fd|: mag = 0 gives inf */
    float travel = 0.0f;
    uint32_t laneSlotBase = 0;
    uint32_t nSamples = 0;
    const float stepSize = f.stepSize;
    int budget = RAYLOD ? vrc_ray_lod_max_hops( f ) /* as vrc_pixel_ray_lod */
                        : 8 * ( f.gridDim[0] + f.gridDim[1] + f.gridDim[2] + 3 ); /* exit guarantee */

    /* staging role of the lane: 4 row-pairs across (x), 16 down (y) per z-slice */
    const uint32_t sxr = lane & 3u, syp = lane >> 2;
    const uint32_t ldsLane = syp * 2u * VRC_LDS_PY + sxr * 8u;
    const uint32_t sliceStride = f.sbx * f.sby * VRC_MB_VOXELS;
    /* classifier of the trilinear form (lds_classify): texel coordinate * 256 + 256.5 from the density * 2^72 */
    [[maybe_unused]] const lds_cls lcls = { cls.mult * 256.0f * 0x1p-72f, cls.add * 256.0f + 256.5f, cls.alphaCorrection };

    /* exit guarantee of the round loop, whatever the state: every round advances at least one lane by a step or a
     * brick, so 64 lanes x (steps of the longest ray + cells it can cross) rounds are never reached */
    uint32_t roundBudget = 64u * ( (uint32_t)( 3.5f / stepSize ) + 8u * (uint32_t)( f.gridDim[0] + f.gridDim[1] + f.gridDim[2] ) + 64u );
    bool events = true, anyLack = true; /* wave-uniform: see the head of the loop */
    uint64_t segMask = 0ull;            /* wave-uniform: the lanes that have a segment (kept as a scalar: in a round
                                         * without events no ballot is spent on it) */
    for( ;; )
    {
        if( roundBudget-- == 0u )
            break;
        /* A: walks.  A lane without a NEXT segment walks its DDA to the next brick it samples -- while it still marches
         * its current one.  The walk (ray/box set-up, ~220 vector instructions) is the whole wave's, so walks are batched:
         * one runs when a lane that lacks a next segment is within VRC_LDS_SOON steps of the end of its current one, or
         * has nothing to march at all (and when VRC_LDS_REFILL lanes lack one: 64 = that trigger is off, see there). */
        auto promote = [&]() {
            if( !done && !hasSeg )
            {
                if( hasPend )
                {
                    fx = pfx; fy = pfy; fz = pfz;
                    fdx = pfdx; fdy = pfdy; fdz = pfdz;
                    /* 1 / |step| per axis for the step counts of a pass: once per segment, not once per pass */
                    rdx = __builtin_amdgcn_rcpf( (float)(uint32_t)( (int32_t)fdx < 0 ? -(int32_t)fdx : (int32_t)fdx ) );
                    rdy = __builtin_amdgcn_rcpf( (float)(uint32_t)( (int32_t)fdy < 0 ? -(int32_t)fdy : (int32_t)fdy ) );
                    rdz = __builtin_amdgcn_rcpf( (float)(uint32_t)( (int32_t)fdz < 0 ? -(int32_t)fdz : (int32_t)fdz ) );
                    travel = pTravel;
                    curNode = pNode;
                    laneSlotBase = pSlotBase;
                    if constexpr( BIG )
                        laneSlotHi = pSlotHi;
                    if constexpr( RAYLOD )
                        lscale = pScale;
                    hasSeg = true;
                    hasPend = false;
                }
                else if( walkDone )
                    done = true;
            }
        };
        /* all of this only when something changed: a segment ended in the last round (events), or lanes are known
         * to lack a next segment (whether they are about to need it changes with every step they take) */
        bool refill = false; /* wave-uniform */
        if( events || anyLack )
        {
            promote();
            const bool lack = !done && !walkDone && !hasPend;
            const uint64_t lackMask = __builtin_amdgcn_ballot_w64( lack );
            const uint64_t idleMask = __builtin_amdgcn_ballot_w64( lack && !hasSeg );
            const uint64_t soonMask = __builtin_amdgcn_ballot_w64(
                lack && hasSeg && !( travel > ( RAYLOD ? stepSize * lscale : stepSize ) * (float)( VRC_LDS_SOON ) ) );
            refill = idleMask != 0ull || soonMask != 0ull || __builtin_popcountll( lackMask ) >= VRC_LDS_REFILL;
            anyLack = lackMask != 0ull;
            if( !refill )
                segMask = __builtin_amdgcn_ballot_w64( hasSeg );
        }
        while( refill )
        {
            VRC_LDS_PHASE( 0 )
            /* the ray and what follows from it are set up again for every walk instead of being kept in ~18
             * registers across the march (same function of the same pixel: same bits); the pixel is passed
             * through an empty asm so that the compiler does not hoist the set-up out of the loop again */
            VRC_LDS_STAT( 3, 1 )
            uint32_t pxw = px, pyw = py;
            asm volatile( "" : "+v"( pxw ), "+v"( pyw ) );
            /* ... and the frame constants the walk needs (the two 4x4 matrices, boxes, planes: ~100 scalars) are read
             * from the kernel-argument segment here, inside the loop, through a pointer the compiler cannot see
             * through: held in scalar registers across the march they spilled into vector lanes by the dozen */
            const __attribute__( ( address_space( 4 ) ) ) vrc_frame* fwp =
                (const __attribute__( ( address_space( 4 ) ) ) vrc_frame*)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile( "" : "+s"( fwp ) );
            const vrc_frame& fw = *(const vrc_frame*)fwp;
            const vrc_ray r = vrc_setup_ray( fw, pxw, fw.rowMap ? fw.rowMap[pyw] : pyw );
            const int stepDir[3] = { r.dir.x > 0.0f ? 1 : -1, r.dir.y > 0.0f ? 1 : -1, r.dir.z > 0.0f ? 1 : -1 };
            const float tDelta[3] = { fw.cellSize[0] * fabsf( r.invDir.x ), fw.cellSize[1] * fabsf( r.invDir.y ),
                                      fw.cellSize[2] * fabsf( r.invDir.z ) };
            if constexpr( RAYLOD )
            {
                if( !done && !walkDone && !hasPend )
                {
                    /* one hop of vrc_pixel_ray_lod (vrc_core.h): the brick at te, marched to where the ray leaves it */
                    if( !( te < t1 ) || --budget < 0 )
                        walkDone = true;
                    else
                    {
                        vrc_dev_node n;
                        float tp, tB;
                        const int32_t node = vrc_ray_lod_hop( fw, r, nodes, gridTable, vrc_ray_lod_base( fw, r ), te, t1, n, tp, tB );
                        if( node >= 0 )
                        {
                            const float scale = (float)( 1u << n.level );
                            const vrc_segment s = vrc_run_segment( r, tp, tB, stepSize * scale );
                            if( s.dist > 0.0f )
                            {
                                const vrc_sampler sm = vrc_make_sampler( n, fw );
                                const vrc_fixpos p0 = vrc_fixpos_init( sm, s.pos, s.step );
                                const uint32_t h = LINEAR ? ( 1u << 23 ) : 0u;
                                pfx = p0.x - h;
                                pfy = p0.y - h;
                                pfz = p0.z - h;
                                pfdx = p0.dx;
                                pfdy = p0.dy;
                                pfdz = p0.dz;
                                pTravel = s.dist;
                                pNode = node;
                                pSlotBase = n.slotBase;
                                if constexpr( BIG )
                                    pSlotHi = n.slotBaseHi;
                                pScale = scale;
                                hasPend = true;
                            }
                        }
                        te = tB;
                    }
                }
            }
            else if( !done && !walkDone && !hasPend )
            {
                if( ddaEnd || --budget < 0 )
                    walkDone = true;
                else
                {
                    /* the walk of vrc_pixel_grid_dda (vrc_core.h), one candidate cell per iteration: the cells
                     * around the edge / corner the ray entered the first cell through or left the last cell
                     * through, then the step through every tied face and the cell behind it */
                    int cx = cell[0], cy = cell[1], cz = cell[2];
                    bool look = false, endAfter = false;
                    if( ( walk & 0x3FFFu ) != 0u )
                    {
                        const uint32_t slot = (uint32_t)__builtin_ctz( walk );
                        walk &= walk - 1u;
                        const uint32_t sub = slot < 7u ? ( 0x4216537u >> ( slot * 4u ) ) & 7u
                                                       : ( 0x653421u >> ( ( slot - 8u ) * 4u ) ) & 7u;
                        const int sgn = slot < 7u ? -1 : 1;
                        cx += ( sub & 1u ) ? sgn * stepDir[0] : 0;
                        cy += ( sub & 2u ) ? sgn * stepDir[1] : 0;
                        cz += ( sub & 4u ) ? sgn * stepDir[2] : 0;
                        look = cx >= 0 && cx < fw.gridDim[0] && cy >= 0 && cy < fw.gridDim[1] && cz >= 0 &&
                               cz < fw.gridDim[2];
                    }
                    else
                    {
                        if( walk & ( 1u << 20 ) )
                        {
                            const uint32_t tied = ( walk >> 16 ) & 7u;
                            if( tied & 1u )
                            {
                                cell[0] += stepDir[0];
                                tMax[0] += tDelta[0];
                            }
                            if( tied & 2u )
                            {
                                cell[1] += stepDir[1];
                                tMax[1] += tDelta[1];
                            }
                            if( tied & 4u )
                            {
                                cell[2] += stepDir[2];
                                tMax[2] += tDelta[2];
                            }
                            walk = 0u;
                            ddaEnd = cell[0] < 0 || cell[0] >= fw.gridDim[0] || cell[1] < 0 ||
                                     cell[1] >= fw.gridDim[1] || cell[2] < 0 || cell[2] >= fw.gridDim[2];
                        }
                        if( !ddaEnd )
                        {
                            cx = cell[0];
                            cy = cell[1];
                            cz = cell[2];
                            look = true;
                            const float tNext = fminf( fminf( tMax[0], tMax[1] ), tMax[2] );
                            if( tNext > t1 )
                                endAfter = true; /* the ray ends inside this cell */
                            else
                            {
                                const float thr = tNext + fabsf( tNext ) * 2e-6f;
                                const uint32_t tied = ( tMax[0] <= thr ? 1u : 0u ) | ( tMax[1] <= thr ? 2u : 0u ) |
                                                      ( tMax[2] <= thr ? 4u : 0u );
                                walk = ( tied << 16 ) | ( 1u << 20 );
                                if( ( tied & ( tied - 1u ) ) != 0u )
                                    walk |= (uint32_t)( ( 0x3F06050003000000ull >> ( tied * 8u ) ) & 0x3Fu ) << 8;
                            }
                        }
                    }
                    const int32_t node = look ? gridTable[( cz * fw.gridDim[1] + cy ) * fw.gridDim[0] + cx] : -1;
                    if( node >= 0 && node != recent0 && node != recent1 && node != recent2 && node != recent3 )
                    {
                        recent3 = recent2;
                        recent2 = recent1;
                        recent1 = recent0;
                        recent0 = node;
                        const vrc_dev_node n = nodes[node];
                        vrc_segment s;
                        bool stop;
                        if( vrc_brick_segment( fw, r, n, stepSize, &s, &stop ) )
                        {
                            if( s.dist > 0.0f )
                            {
                                const vrc_sampler sm = vrc_make_sampler( n, fw );
                                /* same first-sample voxel as the gather kernel */
                                const vrc_fixpos p0 = vrc_fixpos_init( sm, s.pos, s.step );
                                /* trilinear: texel centres at i + 0.5 */
                                const uint32_t h = LINEAR ? ( 1u << 23 ) : 0u;
                                pfx = p0.x - h;
                                pfy = p0.y - h;
                                pfz = p0.z - h;
                                pfdx = p0.dx;
                                pfdy = p0.dy;
                                pfdz = p0.dz;
                                pTravel = s.dist;
                                pNode = node;
                                pSlotBase = n.slotBase;
                                if constexpr( BIG )
                                    pSlotHi = n.slotBaseHi;
                                hasPend = true;
                            }
                        }
                        else if( stop )
                            walkDone = true; /* the reference leaves its brick loop here (Renderer.cu:183-184) */
                    }
                    if( endAfter )
                        ddaEnd = true;
                }
            }
            /* a lane that had nothing to march takes what it found; the walks go on while such lanes remain */
            promote();
            refill = __builtin_amdgcn_ballot_w64( !done && !walkDone && !hasPend && !hasSeg ) != 0ull;
            anyLack = __builtin_amdgcn_ballot_w64( !done && !walkDone && !hasPend ) != 0ull;
            segMask = __builtin_amdgcn_ballot_w64( hasSeg );
        }

        VRC_LDS_PHASE( 6 )
        /* B: one round: every lane that has a segment takes steps through an LDS box (up to VRC_LDS_PASSES boxes);
         * what no box served takes VRC_LDS_G steps by gathers */
        if( segMask == 0ull )
            break;
        /* the lane's step along the ray and its classifier: per-ray LOD scales both with the brick's level */
        const float lstep = RAYLOD ? stepSize * lscale : stepSize;
        [[maybe_unused]] lds_cls lc = lcls;
        if constexpr( RAYLOD )
            lc.kexp = lcls.kexp * lscale;
        bool inTodo = hasSeg;
        bool passEvents = false; /* wave-uniform: a lane crossed the early-exit threshold or ran out of steps in a pass */
        [[maybe_unused]] uint32_t roundSteps = VRC_LDS_G; /* steps the left-over lanes take: as many as the box's lanes took */
        for( int pass = 0; pass < VRC_LDS_PASSES; ++pass )
        {
            const uint64_t todoMask = pass == 0 ? segMask : __builtin_amdgcn_ballot_w64( inTodo );
            if( todoMask == 0ull )
                break;
            /* the lead: the tile centre if it still has steps to take in this round */
            VRC_LDS_PHASE( 1 )
            const uint32_t lead = ( todoMask >> 15 ) & 1ull ? 15u
                                  : ( ( todoMask >> 48 ) & 1ull ? 48u : (uint32_t)__builtin_ctzll( todoMask ) );
            const int32_t brick = __builtin_amdgcn_readlane( curNode, lead );
            const V* slotPtr = atlas + (uint32_t)__builtin_amdgcn_readlane( (int)laneSlotBase, lead );
            if constexpr( BIG )
                slotPtr += (uint64_t)(uint32_t)__builtin_amdgcn_readlane( (int)laneSlotHi, lead ) << 32;
            constexpr uint32_t EXT = LINEAR ? 1u : 0u; /* the taps of a sample reach one voxel further */

            /* ---- the box: where the lanes ARE, extended along the march ------------------------------------
             * Candidates: the lanes of the lead's brick whose current sample lies within 32 voxels around the
             * lead's, per axis.  Three wave-wide ORs (six DPP steps each) of one-hot voxel masks relative to that
             * neighbourhood give the bounding box of their current samples; it is extended in the direction of the
             * march by as many steps (of the lead) as the region holds.  A lane then takes steps until its sample
             * leaves the box (below): lanes that are behind the others in depth -- a lane changes bricks in the
             * middle of a round -- take more steps than those ahead and the tile is level again after the pass,
             * and where the tile is small (10 x 10 voxels across at the front of BASELINE C2's volume against
             * 18 x 18 at its back) the box is deep and a pass is long. */
            const uint32_t ax = fx >> 24, ay = fy >> 24, az = fz >> 24;
            const uint32_t lax = (uint32_t)__builtin_amdgcn_readlane( (int)ax, lead );
            const uint32_t lay = (uint32_t)__builtin_amdgcn_readlane( (int)ay, lead );
            const uint32_t laz = (uint32_t)__builtin_amdgcn_readlane( (int)az, lead );
            const int32_t ldx = __builtin_amdgcn_readlane( (int)fdx, lead ), ldy = __builtin_amdgcn_readlane( (int)fdy, lead ),
                          ldz = __builtin_amdgcn_readlane( (int)fdz, lead );
            const uint32_t wx0 = lax > 15u ? lax - 15u : 0u, wy0 = lay > 15u ? lay - 15u : 0u,
                           wz0 = laz > 15u ? laz - 15u : 0u;
            const bool cand = inTodo && curNode == brick && ax >= wx0 && ax + EXT < wx0 + 32u && ay >= wy0 &&
                              ay + EXT < wy0 + 32u && az >= wz0 && az + EXT < wz0 + 32u;
            const uint32_t one = LINEAR ? 3u : 1u;
            const uint32_t xm = wave_or( cand ? one << ( ax - wx0 ) : 0u );
            const uint32_t ym = wave_or( cand ? one << ( ay - wy0 ) : 0u );
            const uint32_t zm = wave_or( cand ? one << ( az - wz0 ) : 0u );
            /* inclusive voxel ranges of the samples' taps; the lead is a candidate, so no mask is empty */
            uint32_t bx0 = wx0 + (uint32_t)__builtin_ctz( xm ), bx1 = wx0 + 31u - (uint32_t)__builtin_clz( xm );
            uint32_t by0 = wy0 + (uint32_t)__builtin_ctz( ym ), by1 = wy0 + 31u - (uint32_t)__builtin_clz( ym );
            uint32_t bz0 = wz0 + (uint32_t)__builtin_ctz( zm ), bz1 = wz0 + 31u - (uint32_t)__builtin_clz( zm );
            lds_box box;
            {
                /* does a box fit the region: pieces of 8 voxels in x, row pairs in y */
                auto fits = [&]( uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1, uint32_t z0, uint32_t z1 ) {
                    const uint32_t dx = ( x1 | 7u ) - ( x0 & ~7u ) + 1u, dy = ( y1 | 1u ) - ( y0 & ~1u ) + 1u, dz = z1 - z0 + 1u;
                    return dx <= VRC_LDS_PY && dy <= VRC_LDS_MAX_DY && dz <= VRC_LDS_MAX_DZ && dy * dz <= ROWS;
                };
                /* the samples as they are do not fit (lanes far apart): give up the voxels furthest from the
                 * lead's sample, one slab at a time; the lanes that needed them are left for the next pass */
                while( !fits( bx0, bx1, by0, by1, bz0, bz1 ) )
                {
                    const uint32_t s0 = lax - bx0, s1 = bx1 - ( lax + EXT ), s2 = lay - by0, s3 = by1 - ( lay + EXT ),
                                   s4 = laz - bz0, s5 = bz1 - ( laz + EXT );
                    const uint32_t mxy = ( s0 > s1 ? s0 : s1 ) > ( s2 > s3 ? s2 : s3 ) ? ( s0 > s1 ? s0 : s1 ) : ( s2 > s3 ? s2 : s3 );
                    const uint32_t m = mxy > ( s4 > s5 ? s4 : s5 ) ? mxy : ( s4 > s5 ? s4 : s5 );
                    if( m == 0u )
                        break; /* the lead's own taps: 2 x 2 x 2 voxels always fit */
                    if( s0 == m ) ++bx0;
                    else if( s1 == m ) --bx1;
                    else if( s2 == m ) ++by0;
                    else if( s3 == m ) --by1;
                    else if( s4 == m ) ++bz0;
                    else --bz1;
                }
                /* extend by the movement of k steps of the lead (rounded up, plus what the other lanes' slightly
                 * different directions can add), k as large as fits; an axis along which the tile hardly moves
                 * is extended by a voxel on both sides (its lanes may move either way) */
                const uint32_t mx_ = (uint32_t)( ldx < 0 ? -ldx : ldx ), my_ = (uint32_t)( ldy < 0 ? -ldy : ldy ),
                               mz_ = (uint32_t)( ldz < 0 ? -ldz : ldz );
                const uint32_t hx = f.slotDim[0] - 1u, hy = f.slotDim[1] - 1u, hz = f.slotDim[2] - 1u;
                uint32_t ex0 = bx0, ex1 = bx1, ey0 = by0, ey1 = by1, ez0 = bz0, ez1 = bz1;
                for( uint32_t k = VRC_LDS_KMAX; k > 0u; k = k > 8u ? k - 4u : ( k > 4u ? k - 2u : k - 1u ) )
                {
                    auto reach = [&]( uint32_t m ) { return (uint32_t)( ( (uint64_t)m * k + ( m >> 6 ) * k + 0xFFFFFFull ) >> 24 ); };
                    const uint32_t rx = reach( mx_ ), ry = reach( my_ ), rz = reach( mz_ );
                    const bool bothx = mx_ * k < ( 1u << 24 ), bothy = my_ * k < ( 1u << 24 ), bothz = mz_ * k < ( 1u << 24 );
                    const uint32_t nx0 = ( ldx < 0 || bothx ) ? ( bx0 > rx ? bx0 - rx : 0u ) : bx0;
                    const uint32_t nx1 = ( ldx >= 0 || bothx ) ? ( bx1 + rx < hx ? bx1 + rx : hx ) : bx1;
                    const uint32_t ny0 = ( ldy < 0 || bothy ) ? ( by0 > ry ? by0 - ry : 0u ) : by0;
                    const uint32_t ny1 = ( ldy >= 0 || bothy ) ? ( by1 + ry < hy ? by1 + ry : hy ) : by1;
                    const uint32_t nz0 = ( ldz < 0 || bothz ) ? ( bz0 > rz ? bz0 - rz : 0u ) : bz0;
                    const uint32_t nz1 = ( ldz >= 0 || bothz ) ? ( bz1 + rz < hz ? bz1 + rz : hz ) : bz1;
                    if( fits( nx0, nx1, ny0, ny1, nz0, nz1 ) )
                    {
                        ex0 = nx0; ex1 = nx1; ey0 = ny0; ey1 = ny1; ez0 = nz0; ez1 = nz1;
                        break;
                    }
                }
                box.x0 = ex0 & ~7u;
                box.dx = ( ex1 | 7u ) - box.x0 + 1u;
                box.y0 = ey0 & ~1u;
                box.dy = ( ey1 | 1u ) - box.y0 + 1u;
                box.z0 = ez0;
                box.dz = ez1 - ez0 + 1u;
            }
            /* ---- steps a lane takes in this pass: until its sample's taps leave the box -----------------------
             * Per axis the number of steps k >= 0 with lo <= voxel(f + k d) and voxel(f + k d) + EXT <= hi, from a
             * float quotient, then made exact: the last of them must be inside (integer arithmetic, two
             * corrections at most: the quotient is good to a step). */
            uint32_t nSteps = 0, nMin = 0, nMax = 0;
            bool part = false;
            auto countSteps = [&]() {
                const uint32_t lo[3] = { box.x0, box.y0, box.z0 };
                const uint32_t hi[3] = { box.x0 + box.dx - 1u - EXT, box.y0 + box.dy - 1u - EXT, box.z0 + box.dz - 1u - EXT };
                const uint32_t p[3] = { fx, fy, fz }, d[3] = { fdx, fdy, fdz };
                const float rd[3] = { rdx, rdy, rdz };
                float nf = (float)VRC_LDS_NMAX;
                bool inside = cand;
#pragma unroll
                for( int a = 0; a < 3; ++a )
                {
                    const uint32_t v = p[a] >> 24;
                    inside = inside && v >= lo[a] && v <= hi[a];
                    const int32_t sd = (int32_t)d[a];
                    /* distance to the face the lane moves towards, in 2^-24 voxels; voxel hi is left at (hi+1) << 24 */
                    const uint32_t dist = sd < 0 ? p[a] - ( lo[a] << 24 ) : ( ( hi[a] + 1u ) << 24 ) - 1u - p[a];
                    const float q = (float)dist * rd[a]; /* no movement along the axis: inf */
                    nf = fminf( nf, q );
                }
                uint32_t n = inside ? (uint32_t)nf + 1u : 0u; /* steps 0 .. floor(q) */
                n = n > VRC_LDS_NMAX ? VRC_LDS_NMAX : n;
                /* the float quotient is good to a fraction of a step: if the last step it allows is outside after
                 * all, the one before it is inside */
                {
                    const uint32_t k = n > 0u ? n - 1u : 0u;
                    bool ok = true;
#pragma unroll
                    for( int a = 0; a < 3; ++a )
                    {
                        const uint32_t v = ( p[a] + k * d[a] ) >> 24;
                        ok = ok && v >= lo[a] && v <= hi[a];
                    }
                    nSteps = ( n > 1u && !ok ) ? n - 1u : n;
                }
                part = nSteps > 0u;
                /* fewest and most steps of the participating lanes, from one OR of one-hot masks */
                const uint32_t nm = wave_or( part ? 1u << nSteps : 0u );
                nMin = (uint32_t)__builtin_ctz( nm );
                nMax = 31u - (uint32_t)__builtin_clz( nm );
#if !defined( VRC_LDS_NO_STEP_CAP )
                /* a level tile (its lanes within a step of each other) takes whole fast groups and leaves the odd
                 * step to the next pass: the general batches that would take it cost a third of a fast group */
                if( nMax - nMin <= 1u && nMin >= VRC_LDS_GF )
                {
                    nMax = nMin - nMin % VRC_LDS_GF;
                    nSteps = nSteps < nMax ? nSteps : nMax;
                }
#endif
            };
            const uint32_t pz = VRC_LDS_PY * box.dy; /* slice pitch of this pass */
            VRC_LDS_PHASE( 2 )
            /* ---- stage the box: atlas (micro-blocked) -> LDS (linear) --------------------- */
            {
#if defined( VRC_LDS_LOAD_ALL )
                const uint32_t x = box.x0 + ( sxr * 8u < box.dx ? sxr * 8u : box.dx - 8u ),
                               y = box.y0 + ( syp * 2u < box.dy ? syp * 2u : box.dy - 2u );
#else
                const uint32_t x = box.x0 + sxr * 8u, y = box.y0 + syp * 2u;
#endif
                const bool on = sxr * 8u < box.dx && syp * 2u < box.dy;
                const uint32_t partial =
                    ( ( y >> VRC_MB_SHIFT ) * f.sbx + ( x >> VRC_MB_SHIFT ) ) * VRC_MB_VOXELS + vrc_mb_y( y );
                V* const dst = region + ldsLane;
                /* every slice load of a batch is issued before the first LDS write; slices
                 * past the box repeat its last slice (branch-free, same cache lines) */
                /* the lanes' step counts are worked out while the loads are on their way */
                if constexpr( sizeof( V ) == 2 )
                {
                    /* batches of VRC_LDS_STAGE_N16 slices: twice the registers per slice in flight */
                    constexpr uint32_t S = VRC_LDS_STAGE_N16;
                    lds_stage< (int)S >( slotPtr, partial, sliceStride, box.z0, box.dz < S ? box.dz : S, on, dst, pz, countSteps );
                    for( uint32_t z = S; z < box.dz; z += S )
                        lds_stage< (int)S >( slotPtr, partial, sliceStride, box.z0 + z, box.dz - z < S ? box.dz - z : S, on,
                                             dst + z * pz, pz, []() {} );
                }
                else if( box.dz <= 8u )
                    lds_stage< 8 >( slotPtr, partial, sliceStride, box.z0, box.dz, on, dst, pz, countSteps );
                else if( box.dz <= (uint32_t)VRC_LDS_STAGE_N )
                    lds_stage< VRC_LDS_STAGE_N >( slotPtr, partial, sliceStride, box.z0, box.dz, on, dst, pz, countSteps );
                else
                {
                    /* deeper boxes in two halves: at most 11 loads in flight */
                    const uint32_t h = ( box.dz + 1u ) / 2u;
                    lds_stage< VRC_LDS_STAGE_N >( slotPtr, partial, sliceStride, box.z0, h, on, dst, pz, countSteps );
                    lds_stage< VRC_LDS_STAGE_N >( slotPtr, partial, sliceStride, box.z0 + h, box.dz - h, on, dst + h * pz, pz, []() {} );
                }
            }
#if VRC_LDS_GF != VRC_LDS_G
            roundSteps = nMax > roundSteps ? nMax : roundSteps;
#endif
            VRC_LDS_STAT( 0, 1 )
#if defined( VRC_LDS_STATS )
            if( pass == 0 )
            {
                const uint64_t m1_ = __builtin_amdgcn_ballot_w64( inTodo && curNode == brick );
                const uint64_t m4_ = __builtin_amdgcn_ballot_w64( part );
                VRC_LDS_STAT( 1, __builtin_popcountll( todoMask ) )
                VRC_LDS_STAT( 4, __builtin_popcountll( m1_ ) )
#if !defined( VRC_LDS_STATS2 )
                VRC_LDS_STAT( 5, box.dy )
                VRC_LDS_STAT( 6, box.dz )
#endif
                VRC_LDS_STAT( 7, __builtin_popcountll( m4_ ) )
            }
#endif

            /* a wave's LDS accesses complete in issue order, so its own region needs no
             * s_barrier; the compiler must still not move the reads above the copies */
            __builtin_amdgcn_wave_barrier();

            /* ---- march from LDS ------------------------------------------------------------ */
            const uint32_t bias = box.z0 * pz + box.y0 * VRC_LDS_PY + box.x0;
            /* one sample from the box: the entry (rgb*alpha', alpha') it composites */
            auto sampleLds = [&]( uint32_t a, uint32_t sx_, uint32_t sy_, uint32_t sz_ ) -> C {
                if constexpr( LINEAR )
                {
                    float t[8];
                    lds_taps( region + a, region + a + pz, t );
                    return lds_classify( (const C*)nullptr, tab, lds_trilerp( t, sx_, sy_, sz_ ), lc );
                }
                else
                    return lut[(uint32_t)region[a]];
            };
#ifndef VRC_LDS_LBATCH
#define VRC_LDS_LBATCH 2 /* trilinear samples whose 8 taps are read before the first is used */
#endif
            constexpr int BATCH = LINEAR ? VRC_LDS_LBATCH : VRC_LDS_G; /* point sampling: one byte per sample in flight */
            VRC_LDS_PHASE( 3 )
            const C colorIn = color; /* for the replay of a lane that crosses the early-exit threshold */
            uint32_t cnt = 0;        /* samples the lane composited in this pass */
            uint32_t adv = 0;        /* steps its position advanced */
            /* one batch of samples.  FAST: every participating lane takes every step of the batch and its segment
             * goes on beyond it: no per-step selects */
            auto batch = [&]( auto fastTag, uint32_t s0 ) {
                constexpr bool FAST = decltype( fastTag )::value;
                uint32_t a[BATCH], wfx[BATCH], wfy[BATCH], wfz[BATCH];
                bool act[BATCH];
#pragma unroll
                for( int s = 0; s < BATCH; ++s )
                {
                    const bool take = FAST ? true : ( s0 + (uint32_t)s < nSteps );
                    act[s] = FAST ? true : ( take && travel > 0.0f );
                    /* (slice < 256, pitch <= 1024: a 24-bit multiply-add instead of the quarter-rate 32-bit multiply) */
                    const uint32_t av = vrc_mul24( fz >> 24, pz ) + ( ( fy >> 24 ) * VRC_LDS_PY + ( fx >> 24 ) ) - bias;
                    a[s] = take ? av : 0u; /* a step the lane does not take may lie outside the box: it reads offset 0 */
                    wfx[s] = fx;
                    wfy[s] = fy;
                    wfz[s] = fz;
                    fx += take ? fdx : 0u;
                    fy += take ? fdy : 0u;
                    fz += take ? fdz : 0u;
                    /* the positions advance by one addition per step: left alone the compiler turns the eight steps of an
                     * unrolled group into p + k d with the multiples 2d .. 7d set up before every group loop -- eighteen
                     * instructions per pass, six of them quarter-rate 32-bit multiplies, and as many registers */
                    if( FAST )
                        asm volatile( "" : "+v"( fx ), "+v"( fy ), "+v"( fz ) );
                    travel -= take ? lstep : 0.0f;
                    if( !FAST )
                        adv += take ? 1u : 0u;
                }
                C e[BATCH];
                if constexpr( LINEAR )
                {
                    float t[BATCH][8];
#pragma unroll
                    for( int s = 0; s < BATCH; ++s )
                        lds_taps( region + a[s], region + a[s] + pz, t[s] );
#pragma unroll
                    for( int s = 0; s < BATCH; ++s )
                        e[s] = lds_classify( (const C*)nullptr, tab, lds_trilerp( t[s], wfx[s], wfy[s], wfz[s] ), lc );
                }
                else
                {
                    uint32_t d[BATCH];
#pragma unroll
                    for( int s = 0; s < BATCH; ++s )
                        d[s] = (uint32_t)region[a[s]];
#pragma unroll
                    for( int s = 0; s < BATCH; ++s )
                        e[s] = lut[d[s]];
                }
#pragma unroll
                for( int s = 0; s < BATCH; ++s )
                {
                    vrc_composite( color, e[s], !act[s] );
                    if( !FAST )
                        cnt += act[s] ? 1u : 0u;
                }
            };
            if( part ) /* the other lanes keep their state */
            {
                /* groups of VRC_LDS_G steps that every participating lane takes in full, unrolled ... */
                uint32_t s0 = 0;
                while( s0 + VRC_LDS_GF <= nMin &&
                       __builtin_amdgcn_ballot_w64( !( travel > lstep * (float)( VRC_LDS_GF + 1 ) ) ) == 0ull )
                {
#pragma unroll
                    for( int b0 = 0; b0 < VRC_LDS_GF; b0 += BATCH )
                        batch( std::true_type(), 0u );
                    s0 += VRC_LDS_GF;
                    cnt += VRC_LDS_GF;
                    adv += VRC_LDS_GF;
#if defined( VRC_LDS_STATS2 ) /* developer build: unrolled groups / general batches instead of the box shape */
                    VRC_LDS_STAT( 5, 1 )
#endif
                }
                /* ... and the rest, batch by batch, each lane as far as it goes */
#pragma unroll 1
                for( ; s0 < nMax; s0 += BATCH )
                {
                    batch( std::false_type(), s0 );
#if defined( VRC_LDS_STATS2 )
                    VRC_LDS_STAT( 6, 1 )
#endif
                }
                if( COUNT )
                    nSamples += cnt;
            }
            /* early ray termination (Renderer.cu:219-226), tested once per pass: the opacity never decreases, so
             * a lane is over the threshold now iff one of its samples of this pass took it there; such a lane
             * takes its samples again one by one from the colour it came with and stops after the one that
             * crosses -- the reference's exit */
            VRC_LDS_PHASE( 4 )
            const bool crossed = part && color.w > VRC_EARLY_EXIT;
            /* one ballot for "anything happened to a lane in this pass": in most passes nothing did, and neither the
             * replay below nor the end-of-round bookkeeping has to look */
            const bool passEvent = __builtin_amdgcn_ballot_w64( crossed || ( part && !( travel > 0.0f ) ) ) != 0ull;
            passEvents = passEvents || passEvent;
            if( passEvent && __builtin_amdgcn_ballot_w64( crossed ) != 0ull )
            {
                if( crossed )
                {
                    color = colorIn;
                    if( COUNT )
                        nSamples -= cnt;
                    uint32_t rx = fx - adv * fdx, ry = fy - adv * fdy, rz = fz - adv * fdz;
                    bool fin = false;
#pragma unroll 1
                    for( uint32_t s = 0; s < nMax; ++s )
                    {
                        if( !fin && s < cnt )
                        {
                            const uint32_t a = vrc_mul24( rz >> 24, pz ) + ( ( ry >> 24 ) * VRC_LDS_PY + ( rx >> 24 ) ) - bias;
                            vrc_composite( color, sampleLds( a, rx, ry, rz ) );
                            if( COUNT )
                                nSamples += 1u;
                            fin = color.w > VRC_EARLY_EXIT;
                        }
                        rx += fdx;
                        ry += fdy;
                        rz += fdz;
                    }
                    done = true;
                }
            }
            inTodo = inTodo && !part;
            __builtin_amdgcn_wave_barrier(); /* the next pass overwrites the region */
        }

        VRC_LDS_PHASE( 5 )
        /* C: lanes no box of this round served: the same steps by byte gathers from the atlas */
#if defined( VRC_LDS_ABLATE_GATHER ) /* developer timing build: the left-over lanes skip their steps (wrong pixels) */
        if( inTodo )
        {
            fx += 8u * fdx; fy += 8u * fdy; fz += 8u * fdz;
            travel -= 8.0f * lstep;
            inTodo = false;
        }
#endif
        const bool gathers = __builtin_amdgcn_ballot_w64( inTodo ) != 0ull;
        if( gathers )
        {
            VRC_LDS_STAT( 2, 1 )
            if( inTodo )
            {
                const uint32_t cyy = f.sbx * VRC_MB_VOXELS - 64u;
                const uint32_t czz = f.sbx * f.sby * VRC_MB_VOXELS - 512u;
                /* two steps at a time: the sixteen gathers of both are on their way before the first is used */
#ifndef VRC_LDS_GBATCH
#define VRC_LDS_GBATCH 2
#endif
#if VRC_LDS_GF != VRC_LDS_G
#pragma unroll 1
                for( uint32_t s0 = 0; s0 < roundSteps; s0 += VRC_LDS_GBATCH )
#else
#pragma unroll
                for( int s0 = 0; s0 < VRC_LDS_G; s0 += VRC_LDS_GBATCH )
#endif
                {
                    bool act[VRC_LDS_GBATCH];
                    C e[VRC_LDS_GBATCH];
                    if constexpr( LINEAR )
                    {
                        uint32_t wfx[VRC_LDS_GBATCH], wfy[VRC_LDS_GBATCH], wfz[VRC_LDS_GBATCH];
                        float t[VRC_LDS_GBATCH][8];
#pragma unroll
                        for( int s = 0; s < VRC_LDS_GBATCH; ++s )
                        {
                            act[s] = travel > 0.0f;
                            const uint32_t ux = fx >> 24, uy = fy >> 24, uz = fz >> 24;
                            uint32_t ax[2], ay[2], az[2];
#pragma unroll
                            for( int i = 0; i < 2; ++i )
                            {
                                const uint32_t cx = ux + (uint32_t)i, cy = uy + (uint32_t)i, cz = uz + (uint32_t)i;
                                ax[i] = vrc_mul24( cx >> VRC_MB_SHIFT, 504u ) + cx;
                                ay[i] = vrc_mul24( cy >> VRC_MB_SHIFT, cyy ) + ( cy << 3 ) + VRC_MB_FIX_Y( cy );
                                az[i] = vrc_mul24( cz >> VRC_MB_SHIFT, czz ) + ( cz << 6 ) - VRC_MB_FIX_Z( cz ) + ( BIG ? 0u : laneSlotBase );
                            }
#pragma unroll
                            for( int c = 0; c < 8; ++c )
                            {
                                const uint32_t idx = ax[c & 1] + ay[( c >> 1 ) & 1] + az[c >> 2];
                                /* scalar base + 32-bit lane offset form of the load: the offset goes through an empty asm
                                 * so that the instruction selector sees a 32-bit value being widened (a select of
                                 * 64-bit values otherwise: one 64-bit vector addition per gather) */
                                typedef __attribute__( ( address_space( 1 ) ) ) const V lds_g_v;
                                uint32_t off = act[s] ? idx : 0u;
                                asm( "" : "+v"( off ) );
                                if constexpr( BIG ) /* the lane's own slot: a 64-bit base per lane */
                                    t[s][c] = (float)( (lds_g_v*)( atlas + ( ( (uint64_t)laneSlotHi << 32 ) | laneSlotBase ) ) )[off];
                                else
                                    t[s][c] = (float)( (lds_g_v*)atlas )[off];
                            }
                            wfx[s] = fx;
                            wfy[s] = fy;
                            wfz[s] = fz;
                            fx += fdx;
                            fy += fdy;
                            fz += fdz;
                            travel -= lstep;
                        }
#pragma unroll
                        for( int s = 0; s < VRC_LDS_GBATCH; ++s )
                            e[s] = lds_classify( (const C*)nullptr, tab, lds_trilerp( t[s], wfx[s], wfy[s], wfz[s] ), lc );
                    }
                    else
                    {
                        vrc_sampler sm; /* only the address constants are used */
                        sm.slotBase = laneSlotBase;
                        sm.cyy = cyy;
                        sm.czz = czz;
                        uint32_t d[VRC_LDS_GBATCH];
#pragma unroll
                        for( int s = 0; s < VRC_LDS_GBATCH; ++s )
                        {
                            act[s] = travel > 0.0f;
                            const uint32_t iv = vrc_voxel_address( sm, fx >> 24, fy >> 24, fz >> 24 );
                            d[s] = (uint32_t)atlas[act[s] ? iv : 0u];
                            fx += fdx;
                            fy += fdy;
                            fz += fdz;
                            travel -= lstep;
                        }
#pragma unroll
                        for( int s = 0; s < VRC_LDS_GBATCH; ++s )
                            e[s] = lut[act[s] ? d[s] : 256u];
                    }
#pragma unroll
                    for( int s = 0; s < VRC_LDS_GBATCH; ++s )
                    {
                        const bool on = act[s] && !done;
                        vrc_composite( color, e[s], !on );
                        if( COUNT )
                            nSamples += on ? 1u : 0u;
                        done = done || ( on && color.w > VRC_EARLY_EXIT );
                    }
                }
            }
        }
        events = false;
        if( passEvents || gathers )
        {
            const bool ended = hasSeg && ( done || !( travel > 0.0f ) );
            const uint64_t endedMask = __builtin_amdgcn_ballot_w64( ended );
            events = endedMask != 0ull;
            segMask &= ~endedMask;
            if( ended )
                hasSeg = false;
        }
        VRC_LDS_PHASE( 6 )
    }

#if defined( VRC_LDS_TIMING )
    VRC_LDS_PHASE( 7 )
    if( lane == 0 )
        for( int i = 0; i < 8; ++i )
            atomicAdd( &vrc_lds_phase[i], phaseAcc[i] );
#endif
    if( store )
    {
        if constexpr( GREY )
            pixelBuffer[pixelPos] = vrc_f4{ color.x, color.x, color.x, color.w };
        else
            pixelBuffer[pixelPos] = color;
    }
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

namespace
{
template < bool LINEAR, bool RAYLOD, typename V, bool BIG = false >
hipError_t launch_lds( const vrc_raycast_args& a, hipStream_t stream )
{
    const uint32_t tilesX = ( a.frame.width + 7u ) / 8u;
    const uint32_t tilesY = ( a.frame.height + 7u ) / 8u;
    const uint32_t nTiles = tilesX * tilesY;
    const dim3 grid( ( vrc_schedule_slots( tilesX, tilesY ) + VRC_LDS_WAVES - 1u ) / VRC_LDS_WAVES ),
        block( 64u * VRC_LDS_WAVES );
    const bool count = a.sampleCounter != nullptr;
    /* the instance as rocprofv3 prints it (defaulted template arguments are printed too) */
    vrc_internal_note_kernel( "vrc_k_raycast_lds<%s,%s,%s,%s,%s,%s>", count ? "true" : "false", LINEAR ? "true" : "false",
                              a.greyTable ? "true" : "false", RAYLOD ? "true" : "false",
                              sizeof( V ) == 1 ? "unsigned char" : "unsigned short", BIG ? "true" : "false" );
#define VRC_LDS_LAUNCH( COUNT, GREY )                                                                             \
    hipLaunchKernelGGL( ( vrc_k_raycast_lds< COUNT, LINEAR, GREY, RAYLOD, V, BIG > ), grid, block, 0, stream, a.frame, \
                        a.nodes, a.gridTable, (const V*)a.atlas, a.lut, a.classifier, a.pixelBuffer,              \
                        a.sampleCounter, a.tileOrder, tilesX, nTiles )
    if( a.greyTable )
    {
        if( count ) VRC_LDS_LAUNCH( true, true ); else VRC_LDS_LAUNCH( false, true );
    }
    else
    {
        if( count ) VRC_LDS_LAUNCH( true, false ); else VRC_LDS_LAUNCH( false, false );
    }
#undef VRC_LDS_LAUNCH
    return hipGetLastError();
}
}

hipError_t vrc_launch_raycast_lds( const vrc_raycast_args& a, hipStream_t stream )
{
    if( a.frame.width == 0u || a.frame.height == 0u )
        return hipSuccess;
    const bool rayLod = a.frame.lodLevels > 0u; /* set by the host for vrc_set_ray_lod frames only */
    /* point sampling reads the 257-entry classified table: 8-bit voxels, one level */
    /* atlases of more than 2^32 voxels: the trilinear form without per-ray LOD */
    if( ( a.elemBytes != 1u && a.elemBytes != 2u ) || ( !a.linear && ( rayLod || a.elemBytes != 1u ) ) ||
        ( a.bigAtlas && ( !a.linear || rayLod ) ) || a.clamp || !a.gridTable )
        return hipErrorInvalidValue;
    if( rayLod && ( a.frame.variant != VRC_VARIANT_CUDA || a.frame.lodLevels > VRC_MAX_LOD_LEVELS ) )
        return hipErrorInvalidValue;
    if( !a.linear )
        return launch_lds< false, false, uint8_t >( a, stream );
    if( a.bigAtlas )
        return a.elemBytes == 2u ? launch_lds< true, false, uint16_t, true >( a, stream )
                                 : launch_lds< true, false, uint8_t, true >( a, stream );
    if( a.elemBytes == 2u )
        return rayLod ? launch_lds< true, true, uint16_t >( a, stream ) : launch_lds< true, false, uint16_t >( a, stream );
    return rayLod ? launch_lds< true, true, uint8_t >( a, stream ) : launch_lds< true, false, uint8_t >( a, stream );
}

/*
 * vrc_kernels_lds.hip -- LDS-staged form of the raycast kernel (gfx950).
 *
 * Same integrator and the same sample sequence as vrc_k_raycast (cuda/Renderer.cu:95-230),
 * but the voxels a wave needs for the next few steps are first copied into LDS with wide
 * coalesced loads and the march then reads LDS instead of issuing byte gathers:
 *
 *   - one wave64 = one 8x8 pixel tile (as before); a workgroup is VRC_LDS_WAVES independent
 *     waves that only share the 4 KiB transfer-function table;
 *   - the wave works on ONE brick at a time: lanes whose current brick is another one wait
 *     (a tile covers at most 2x2 bricks across, usually one);
 *   - a round = up to VRC_LDS_G steps of every participating lane.  The slot-local bounding
 *     box of all the voxels those samples touch is found with six DPP wave reductions, the
 *     box is copied atlas -> LDS as 16-byte pieces (two 8-voxel rows of a micro-block slice,
 *     one global_load_dwordx4 + one ds_write2_b64 per lane and z-slice), and the samples are
 *     taken from LDS with a linear address (z*PZ + y*PY + x, compile-time pitches, so the
 *     eight trilinear taps are immediate offsets of one address);
 *   - the region comes in two shapes (flat 32x24x11; deeper 32x20x16 for views along y);
 *   - a box that does not fit the LDS region halves the round (8,4,2,1 steps) and then the
 *     lane set (half tile, quarter tile, 2x2 quad, single lane), so any view is handled.
 *
 * Why: a 64-lane byte gather occupies the CU's texture addresser for ~23 cycles whatever it
 * returns.  Point sampling needs one gather per sample and the staging
 * overhead (reductions + copies, ~100 VALU per round) does not pay; the trilinear filter
 * needs eight taps per sample and does: this kernel is the fast path of VRC_OPT_FILTER = 1
 * and a measured alternative (VRC_OPT_KERNEL = VRC_KERNEL_LDS) for point sampling.
 *
 * Requires overlap >= 1 (no clamped sampler), slot dims <= 248 and a grid-aligned node set;
 * vrc_api.hip falls back to the gather kernels otherwise.
 */
#include "vrc_internal.h"

#include <cstdlib>
#include <type_traits>

/* LDS region of one wave: PY = row pitch = max x extent (a multiple of 8), RY / RZ = max y / z
 * extent, bytes = PY*RY*RZ.  Two shapes, picked per frame from the view direction in volume space
 * (vrc_launch_raycast_lds): the flat one by default, the deeper one for views within ~20 degrees
 * of the y axis, the only ones it helps (measured, C2 trilinear, ms per frame flat / deep: along z
 * 2.46 / 2.71, along y 3.36 / 3.07, along x 3.59 / 3.72, 30/20 degrees off axis 3.59 / 4.17). */
struct vrc_lds_shape_flat
{
    static constexpr uint32_t PY = 32u, RY = 24u, RZ = 11u, PZ = PY * RY, REGION = PZ * RZ;
};
struct vrc_lds_shape_deep
{
    static constexpr uint32_t PY = 32u, RY = 20u, RZ = 16u, PZ = PY * RY, REGION = PZ * RZ;
};
#ifndef VRC_LDS_WAVES
#define VRC_LDS_WAVES 4u
#endif
#ifndef VRC_LDS_G
#define VRC_LDS_G 8
#endif
#ifndef VRC_LDS_REFILL
#define VRC_LDS_REFILL 8
#endif

#if defined( VRC_LDS_STATS ) /* developer build only (tools/build_variants.sh) */
__device__ unsigned long long vrc_lds_stats[8];
__device__ unsigned int vrc_lds_log[64 * 16];
__device__ unsigned int vrc_lds_log_n;
extern "C" int vrc_debug_lds_log( unsigned int out[64 * 16] )
{
    return hipMemcpyFromSymbol( out, HIP_SYMBOL( vrc_lds_log ), sizeof( unsigned int ) * 64 * 16 ) == hipSuccess ? 0 : 1;
}
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
#define VRC_LDS_STAT( I, V )
#endif

namespace
{
/* wave64 min / max: four DPP steps inside each row of 16 lanes, two row broadcasts, result
 * read from lane 63 */
template < bool MAX >
__device__ __forceinline__ uint32_t wave_reduce( uint32_t v )
{
#define VRC_DPP_STEP( CTRL, ROWMASK )                                                          \
    {                                                                                          \
        const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp( (int)v, (int)v, CTRL,        \
                                                                  ROWMASK, 0xF, false );       \
        v = MAX ? ( t > v ? t : v ) : ( t < v ? t : v );                                       \
    }
    VRC_DPP_STEP( 0xB1, 0xF )  /* quad_perm [1,0,3,2] */
    VRC_DPP_STEP( 0x4E, 0xF )  /* quad_perm [2,3,0,1] */
    VRC_DPP_STEP( 0x141, 0xF ) /* row_half_mirror */
    VRC_DPP_STEP( 0x140, 0xF ) /* row_mirror */
    VRC_DPP_STEP( 0x142, 0xA ) /* row_bcast:15 into rows 1 and 3 */
    VRC_DPP_STEP( 0x143, 0xC ) /* row_bcast:31 into rows 2 and 3 */
#undef VRC_DPP_STEP
    return (uint32_t)__builtin_amdgcn_readlane( (int)v, 63 );
}

/* copy N z-slices of the box: per lane one 16-byte piece (two 8-voxel rows) per slice */
template < typename S, int N >
__device__ __forceinline__ void lds_stage( const uint8_t* __restrict__ slotPtr, uint32_t partial,
                                           uint32_t sliceStride, uint32_t z0, uint32_t dz, bool on,
                                           uint8_t* dst )
{
    if( !on )
        return;
    uint4 v[N];
#pragma unroll
    for( int z = 0; z < N; ++z )
    {
        const uint32_t zc = (uint32_t)z < dz ? (uint32_t)z : dz - 1u;
        const uint32_t zz = z0 + zc;
        const uint8_t* const zb = slotPtr + ( ( zz >> VRC_MB_SHIFT ) * sliceStride + ( ( zz & 7u ) << 6 ) );
        v[z] = *reinterpret_cast< const uint4* >( zb + partial );
    }
#pragma unroll
    for( int z = 0; z < N; ++z )
    {
        if( (uint32_t)z < dz ) /* wave-uniform; never write past the box (the region ends with it) */
        {
            *reinterpret_cast< uint2* >( dst + z * S::PZ ) = make_uint2( v[z].x, v[z].y );
            *reinterpret_cast< uint2* >( dst + z * S::PZ + S::PY ) = make_uint2( v[z].z, v[z].w );
        }
    }
}

struct lds_box
{
    uint32_t x0, y0, z0; /* origin: x0 multiple of 8, y0 even */
    uint32_t dx, dy, dz; /* extents */
};
}

/* GREY: the transfer function is grey and the frame starts from zero (vrc_raycast_args.greyTable): colours and table
 * entries are (grey, alpha) pairs, bit-identical to the four-float form (vrc_core.h, VRC_MODE_GREY) */
template < bool COUNT, bool LINEAR, typename S, bool GREY = false >
/* the deep region leaves LDS for three workgroups per CU (3 waves per SIMD), the flat one for four */
#ifndef VRC_LDS_LINEAR_WAVES
#define VRC_LDS_LINEAR_WAVES 4 /* measured: 4 waves beat 3 (2.52 against 2.71 ms, when 4 still spilled 11 dwords) */
#endif
__global__ __launch_bounds__( 64 * VRC_LDS_WAVES, S::REGION > 8448u ? 3 : ( LINEAR ? VRC_LDS_LINEAR_WAVES : 4 ) ) void vrc_k_raycast_lds(
    const vrc_frame f, const vrc_dev_node* __restrict__ nodes,
    const int32_t* __restrict__ gridTable, const uint8_t* __restrict__ atlas,
    const vrc_f4* __restrict__ lutGlobal, const vrc_classifier cls,
    vrc_f4* __restrict__ pixelBuffer, unsigned long long* __restrict__ sampleCounter,
    const uint32_t* __restrict__ tileOrder, const uint32_t tilesX, const uint32_t nTiles )
{
    using C = std::conditional_t< GREY, vrc_f2, vrc_f4 >; /* a colour / a table entry */
    __shared__ C lut[VRC_TFP_ENTRIES];
    __shared__ __attribute__( ( aligned( 16 ) ) ) uint8_t regions[VRC_LDS_WAVES][S::REGION];

    for( uint32_t i = threadIdx.x; i < VRC_TFP_ENTRIES; i += 64u * VRC_LDS_WAVES )
    {
        const vrc_f4 e = lutGlobal[i];
        if constexpr( GREY )
            lut[i] = vrc_f2{ e.x, e.w };
        else
            lut[i] = e;
    }
    __syncthreads();
    /* from here on the waves of the workgroup are independent: no further barrier */

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t slot = blockIdx.x * VRC_LDS_WAVES + wave;
    const uint32_t tilesY = nTiles / tilesX;
    if( slot >= vrc_schedule_slots( tilesX, tilesY ) )
        return;
    uint8_t* const region = regions[wave];

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

    int cell[3] = { 0, 0, 0 }, stepDir[3] = { 1, 1, 1 };
    float tMax[3] = { 0.f, 0.f, 0.f }, tDelta[3] = { 0.f, 0.f, 0.f };
    float t0 = 0.0f, t1 = 0.0f;
    uint32_t back = 0u;
    bool ddaEnd = false;
    if( !done )
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
                stepDir[a] = pos ? 1 : -1;
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
                tDelta[a] = f.cellSize[a] * fabsf( id[a] );
            }
        }
    }

    /* ---- per-lane segment state ---------------------------------------------------------- */
    bool hasSeg = false;
    int32_t curNode = -1;
    /* bricks already handed to the slab test; probe: what the walk does next at its cell (0: the cell
     * itself, 1..6: the cells around an edge / corner the ray leaves through, see vrc_pixel_grid_dda) */
    int32_t recent0 = -1, recent1 = -1, recent2 = -1, recent3 = -1;
    /* work list of cells around an edge / corner (encoding: vrc_pixel_grid_dda), the faces the ray leaves
     * the current cell through, and whether the step through them is still to be taken */
    /* bits 0-13: the work list; bits 16-18: tied axes; bit 20: the step through them is still to be taken
     * (one register: the trilinear form is short of them) */
    uint32_t walk = (uint32_t)( ( 0x7F68544032201000ull >> ( back * 8u ) ) & 0x7Fu );
    uint32_t fx = 0, fy = 0, fz = 0, fdx = 0, fdy = 0, fdz = 0; /* 8.24 slot-local voxel */
    float travel = 0.0f;
    uint32_t laneSlotBase = 0;
    uint32_t nSamples = 0;
    const float stepSize = f.stepSize;
    const float invStep = 1.0f / stepSize;
    int budget = 8 * ( f.gridDim[0] + f.gridDim[1] + f.gridDim[2] + 3 ); /* exit guarantee */

    /* staging role of the lane: 4 row-pairs across (x), 16 down (y) per z-slice */
    const uint32_t sxr = lane & 3u, syp = lane >> 2;
    const uint32_t ldsLane = syp * 2u * S::PY + sxr * 8u;
    const uint32_t sliceStride = f.sbx * f.sby * VRC_MB_VOXELS;

    for( ;; )
    {
        /* A: live lanes without a segment walk their DDA to the next brick they sample.  The
         * walk (ray/box set-up, ~200 instructions) is shared by the wave, so lanes that
         * finish a brick early wait for company (VRC_LDS_REFILL lanes) or for the others to
         * run dry, as the lanes of the gather kernel wait at the end of a brick's march loop. */
        bool refill; /* wave-uniform */
        {
            const uint64_t needMask = __builtin_amdgcn_ballot_w64( !done && !hasSeg );
            const uint64_t haveMask = __builtin_amdgcn_ballot_w64( hasSeg );
            /* lanes that leave their brick within the next round: waiting for them keeps the
             * wave in step (lanes that enter a brick one round apart are 8 voxels apart in
             * depth for the rest of it and never share a box again) */
            const uint64_t soonMask = __builtin_amdgcn_ballot_w64(
                hasSeg && !( travel > stepSize * (float)VRC_LDS_G ) );
            refill = !( needMask != 0ull && haveMask != 0ull &&
                        ( __builtin_popcountll( needMask ) < VRC_LDS_REFILL || soonMask != 0ull ) );
        }
        while( refill && __builtin_amdgcn_ballot_w64( !done && !hasSeg ) != 0ull )
        {
            VRC_LDS_STAT( 4, 1 )
            if( !done && !hasSeg )
            {
                if( ddaEnd || --budget < 0 )
                    done = true;
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
                        look = cx >= 0 && cx < f.gridDim[0] && cy >= 0 && cy < f.gridDim[1] && cz >= 0 &&
                               cz < f.gridDim[2];
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
                            ddaEnd = cell[0] < 0 || cell[0] >= f.gridDim[0] || cell[1] < 0 ||
                                     cell[1] >= f.gridDim[1] || cell[2] < 0 || cell[2] >= f.gridDim[2];
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
                    const int32_t node = look ? gridTable[( cz * f.gridDim[1] + cy ) * f.gridDim[0] + cx] : -1;
                    if( node >= 0 && node != recent0 && node != recent1 && node != recent2 && node != recent3 )
                    {
                        recent3 = recent2;
                        recent2 = recent1;
                        recent1 = recent0;
                        recent0 = node;
                        const vrc_dev_node n = nodes[node];
                        vrc_segment s;
                        bool stop;
                        if( vrc_brick_segment( f, r, n, stepSize, &s, &stop ) )
                        {
                            if( s.dist > 0.0f )
                            {
                                const vrc_sampler sm = vrc_make_sampler( n, f );
                                /* same first-sample voxel as the gather kernel */
                                const vrc_fixpos p0 = vrc_fixpos_init( sm, s.pos, s.step );
                                /* trilinear: texel centres at i + 0.5 */
                                const uint32_t h = LINEAR ? ( 1u << 23 ) : 0u;
                                fx = p0.x - h;
                                fy = p0.y - h;
                                fz = p0.z - h;
                                fdx = p0.dx;
                                fdy = p0.dy;
                                fdz = p0.dz;
                                travel = s.dist;
                                curNode = node;
                                laneSlotBase = n.slotBase;
                                hasSeg = true;
                            }
                        }
                        else if( stop )
                            done = true;
                    }
                    if( endAfter )
                        ddaEnd = true;
                }
            }
        }

        /* B: the round's LDS box is built around one lane (the tile centre if it has a
         * segment): its brick, and the lanes of that brick that are near it */
        const uint64_t segMask = __builtin_amdgcn_ballot_w64( hasSeg );
        if( segMask == 0ull )
            break;
        const uint32_t lead = ( segMask >> 15 ) & 1ull ? 15u
                              : ( ( segMask >> 48 ) & 1ull ? 48u : (uint32_t)__builtin_ctzll( segMask ) );
        const int32_t brick = __builtin_amdgcn_readlane( curNode, lead );
        const uint8_t* const slotPtr = atlas + (uint32_t)__builtin_amdgcn_readlane( (int)laneSlotBase, lead );

        /* C: one round: up to g steps of every lane that has a segment -- from LDS for the
         * lanes in the box, by byte gathers from the atlas for the others */
        {
            bool part = hasSeg && curNode == brick;
            int g = VRC_LDS_G;
            uint32_t lox, hix, loy, hiy, loz, hiz;
            /* voxels touched by the lane's next (up to) g samples; the step count is
             * over-estimated by at most 2 (the exact per-sample test is travel > 0).  g is
             * halved until the first participating lane's own footprint fits the region. */
            for( ;; )
            {
                int m = (int)( travel * invStep ) + 2;
                m = m < g ? m : g;
                const uint32_t k = (uint32_t)( m - 1 );
                const uint32_t ex = fx + k * fdx, ey = fy + k * fdy, ez = fz + k * fdz;
                const uint32_t ax = fx >> 24, ay = fy >> 24, az = fz >> 24;
                const uint32_t bx = ex >> 24, by = ey >> 24, bz = ez >> 24;
                const uint32_t ext = LINEAR ? 1u : 0u;
                lox = ax < bx ? ax : bx; hix = ( ax < bx ? bx : ax ) + ext;
                loy = ay < by ? ay : by; hiy = ( ay < by ? by : ay ) + ext;
                loz = az < bz ? az : bz; hiz = ( az < bz ? bz : az ) + ext;
                /* the over-estimate may leave the slot: keep the copy inside it */
                hix = hix < f.slotDim[0] - 1u ? hix : f.slotDim[0] - 1u;
                hiy = hiy < f.slotDim[1] - 1u ? hiy : f.slotDim[1] - 1u;
                hiz = hiz < f.slotDim[2] - 1u ? hiz : f.slotDim[2] - 1u;
                lox = lox < hix ? lox : hix;
                loy = loy < hiy ? loy : hiy;
                loz = loz < hiz ? loz : hiz;
                /* x origin is aligned down to 8, y origin to 2 */
                const bool ownFit = hix - ( lox & ~7u ) < S::PY && hiy - ( loy & ~1u ) < S::RY &&
                                    hiz - loz < S::RZ;
                const bool leadFit = ( __builtin_amdgcn_ballot_w64( ownFit ) >> lead ) & 1ull;
                if( leadFit || g == 1 )
                {
                    part = part && ownFit;
                    break;
                }
                g >>= 1;
            }
            lds_box box;
            bool windowed = false;
            for( ;; )
            {
                const uint32_t mnx = wave_reduce< false >( part ? lox : 0xFFFFFFFFu );
                const uint32_t mny = wave_reduce< false >( part ? loy : 0xFFFFFFFFu );
                const uint32_t mnz = wave_reduce< false >( part ? loz : 0xFFFFFFFFu );
                const uint32_t mxx = wave_reduce< true >( part ? hix : 0u );
                const uint32_t mxy = wave_reduce< true >( part ? hiy : 0u );
                const uint32_t mxz = wave_reduce< true >( part ? hiz : 0u );
                box.x0 = mnx & ~7u;
                box.y0 = mny & ~1u;
                box.z0 = mnz;
                box.dx = mxx - box.x0 + 1u;
                box.dy = mxy - box.y0 + 1u;
                box.dz = mxz - box.z0 + 1u;
                VRC_LDS_STAT( 3, 1 )
                if( ( box.dx <= S::PY && box.dy <= S::RY && box.dz <= S::RZ ) || windowed )
                    break;
                /* The lanes of the brick are too far apart (rays that entered it through
                 * different faces are at different depths): keep the lanes whose footprint
                 * lies in a region-sized window centred on the first lane's footprint; the
                 * others are served by later rounds. */
                const uint32_t llx = (uint32_t)__builtin_amdgcn_readlane( (int)lox, lead );
                const uint32_t lhx = (uint32_t)__builtin_amdgcn_readlane( (int)hix, lead );
                const uint32_t lly = (uint32_t)__builtin_amdgcn_readlane( (int)loy, lead );
                const uint32_t lhy = (uint32_t)__builtin_amdgcn_readlane( (int)hiy, lead );
                const uint32_t llz = (uint32_t)__builtin_amdgcn_readlane( (int)loz, lead );
                const uint32_t lhz = (uint32_t)__builtin_amdgcn_readlane( (int)hiz, lead );
                const uint32_t ex_ = lhx - llx + 1u, ey_ = lhy - lly + 1u, ez_ = lhz - llz + 1u;
                const uint32_t sx = ex_ < S::PY - 7u ? ( S::PY - 7u - ex_ ) / 2u : 0u;
                const uint32_t sy = ey_ < S::RY - 1u ? ( S::RY - 1u - ey_ ) / 2u : 0u;
                const uint32_t sz = ez_ < S::RZ ? ( S::RZ - ez_ ) / 2u : 0u;
                const uint32_t wx0 = ( llx - ( llx < sx ? llx : sx ) ) & ~7u;
                const uint32_t wy0 = ( lly - ( lly < sy ? lly : sy ) ) & ~1u;
                const uint32_t wz0 = llz - ( llz < sz ? llz : sz );
                part = part && lox >= wx0 && hix < wx0 + S::PY && loy >= wy0 &&
                       hiy < wy0 + S::RY && loz >= wz0 && hiz < wz0 + S::RZ;
                windowed = true;
            }

            VRC_LDS_STAT( 0, 1 )
            VRC_LDS_STAT( 1, g )
            {
                const uint64_t pm_ = __builtin_amdgcn_ballot_w64( part );
                (void)pm_;
                VRC_LDS_STAT( 5, __builtin_popcountll( pm_ ) )
            }
            VRC_LDS_STAT( 6, box.dx * box.dy * box.dz )
            VRC_LDS_STAT( 7, box.dz )
            /* ---- stage the box: atlas (micro-blocked) -> LDS (linear) --------------------- */
            {
                const uint32_t x = box.x0 + sxr * 8u, y = box.y0 + syp * 2u;
                const bool on = sxr * 8u < box.dx && syp * 2u < box.dy;
                const uint32_t partial =
                    ( ( y >> VRC_MB_SHIFT ) * f.sbx + ( x >> VRC_MB_SHIFT ) ) * VRC_MB_VOXELS +
                    ( ( y & 7u ) << 3 );
                uint8_t* const dst = region + ldsLane;
                /* every slice load of the round is issued before the first LDS write; slices
                 * past the box repeat its last slice (branch-free, same cache lines) */
                if( box.dz <= 8u )
                    lds_stage< S, 8 >( slotPtr, partial, sliceStride, box.z0, box.dz, on, dst );
                else if( box.dz <= 11u )
                    lds_stage< S, 11 >( slotPtr, partial, sliceStride, box.z0, box.dz, on, dst );
                else
                {
                    /* deep boxes (S::RZ > 11) in two halves: at most 11 loads in flight */
                    const uint32_t h = ( box.dz + 1u ) / 2u;
                    lds_stage< S, 11 >( slotPtr, partial, sliceStride, box.z0, h, on, dst );
                    lds_stage< S, 11 >( slotPtr, partial, sliceStride, box.z0 + h, box.dz - h, on, dst + h * S::PZ );
                }
            }
            /* a wave's LDS accesses complete in issue order, so its own region needs no
             * s_barrier; the compiler must still not move the reads above the copies */
            __builtin_amdgcn_wave_barrier();

            /* ---- march g steps from LDS ---------------------------------------------------- */
            const uint32_t bias = box.z0 * S::PZ + box.y0 * S::PY + box.x0;
#ifndef VRC_LDS_LBATCH
#define VRC_LDS_LBATCH 2 /* trilinear samples whose 8 taps are read before the first is used: 4 spills at four waves per SIMD (2.39 -> 2.35 ms with 2; 8: 4.3 ms) */
#endif
            constexpr int BATCH = LINEAR ? VRC_LDS_LBATCH : VRC_LDS_G;
            /* FASTR: a full round (g = VRC_LDS_G) in which every participating lane has more than
             * g steps left: no per-step "does this lane take this step" selects */
            auto marchLds = [&]( auto fastTag ) {
                constexpr bool FASTR = decltype( fastTag )::value;
#pragma unroll
                for( int b0 = 0; b0 < VRC_LDS_G; b0 += BATCH )
                {
                    if( FASTR || b0 < g ) /* wave-uniform */
                    {
                        /* addresses of the batch; a step the lane does not take reads offset 0 */
                        uint32_t a[BATCH], wfx[BATCH], wfy[BATCH], wfz[BATCH];
                        bool act[BATCH];
#pragma unroll
                        for( int s = 0; s < BATCH; ++s )
                        {
                            const bool take = FASTR ? true : ( part && ( b0 + s < g ) );
                            act[s] = FASTR ? true : ( take && travel > 0.0f );
                            const uint32_t av = ( fz >> 24 ) * S::PZ + ( fy >> 24 ) * S::PY +
                                                ( fx >> 24 ) - bias;
                            a[s] = act[s] ? av : 0u;
                            wfx[s] = fx;
                            wfy[s] = fy;
                            wfz[s] = fz;
                            fx += take ? fdx : 0u;
                            fy += take ? fdy : 0u;
                            fz += take ? fdz : 0u;
                            travel -= take ? stepSize : 0.0f;
                        }
                        C e[BATCH];
                        if( LINEAR )
                        {
                            float t[BATCH][8];
#pragma unroll
                            for( int s = 0; s < BATCH; ++s )
                            {
                                const uint8_t* const p = region + a[s];
                                t[s][0] = (float)p[0];
                                t[s][1] = (float)p[1];
                                t[s][2] = (float)p[S::PY];
                                t[s][3] = (float)p[S::PY + 1u];
                                t[s][4] = (float)p[S::PZ];
                                t[s][5] = (float)p[S::PZ + 1u];
                                t[s][6] = (float)p[S::PZ + S::PY];
                                t[s][7] = (float)p[S::PZ + S::PY + 1u];
                            }
#pragma unroll
                            for( int s = 0; s < BATCH; ++s )
                            {
                                const float sc = 1.0f / 16777216.0f;
                                const float wx = (float)( wfx[s] & 0xFFFFFFu ) * sc;
                                const float wy = (float)( wfy[s] & 0xFFFFFFu ) * sc;
                                const float wz = (float)( wfz[s] & 0xFFFFFFu ) * sc;
                                e[s] = vrc_classify( lut, vrc_trilerp( t[s], wx, wy, wz ), cls );
                            }
                        }
                        else
                        {
                            uint32_t d[BATCH];
#pragma unroll
                            for( int s = 0; s < BATCH; ++s )
                                d[s] = (uint32_t)region[a[s]];
#pragma unroll
                            for( int s = 0; s < BATCH; ++s )
                                e[s] = lut[act[s] ? d[s] : 256u];
                        }
#pragma unroll
                        for( int s = 0; s < BATCH; ++s )
                        {
                            const bool on = act[s] && !done;
                            vrc_composite( color, e[s], !on );
                            if( COUNT )
                                nSamples += on ? 1u : 0u;
                            done = done || ( on && color.w > VRC_EARLY_EXIT );
                        }
                    }
                }
            };
            const bool fastRound =
                g == VRC_LDS_G &&
                __builtin_amdgcn_ballot_w64( part && !( travel > stepSize * (float)( VRC_LDS_G + 1 ) ) ) == 0ull;
            if( fastRound )
            {
                if( part ) /* the lanes outside the box keep their state */
                    marchLds( std::true_type() );
            }
            else
                marchLds( std::false_type() );
            /* ---- the other lanes with a segment: same steps by gathers from the atlas ------ */
            const bool strag = hasSeg && !part;
            if( __builtin_amdgcn_ballot_w64( strag ) != 0ull )
            {
                VRC_LDS_STAT( 2, 1 )
                vrc_sampler sm; /* only the address constants are used */
                sm.slotBase = laneSlotBase;
                sm.cyy = f.sbx * VRC_MB_VOXELS - 64u;
                sm.czz = f.sbx * f.sby * VRC_MB_VOXELS - 512u;
#pragma unroll
                for( int b0 = 0; b0 < VRC_LDS_G; b0 += BATCH )
                {
                    if( b0 < g )
                    {
                        bool act[BATCH];
                        C e[BATCH];
                        if( LINEAR )
                        {
                            uint32_t ax[BATCH][2], ay[BATCH][2], az[BATCH][2], wfx[BATCH], wfy[BATCH], wfz[BATCH];
#pragma unroll
                            for( int s = 0; s < BATCH; ++s )
                            {
                                const bool take = strag && ( b0 + s < g );
                                act[s] = take && travel > 0.0f;
                                const uint32_t ux = fx >> 24, uy = fy >> 24, uz = fz >> 24;
#pragma unroll
                                for( int i = 0; i < 2; ++i )
                                {
                                    const uint32_t cx = ux + (uint32_t)i, cy = uy + (uint32_t)i, cz = uz + (uint32_t)i;
                                    ax[s][i] = vrc_mul24( cx >> VRC_MB_SHIFT, 504u ) + cx;
                                    ay[s][i] = vrc_mul24( cy >> VRC_MB_SHIFT, sm.cyy ) + ( cy << 3 );
                                    az[s][i] = vrc_mul24( cz >> VRC_MB_SHIFT, sm.czz ) + ( cz << 6 ) + sm.slotBase;
                                }
                                wfx[s] = fx;
                                wfy[s] = fy;
                                wfz[s] = fz;
                                fx += take ? fdx : 0u;
                                fy += take ? fdy : 0u;
                                fz += take ? fdz : 0u;
                                travel -= take ? stepSize : 0.0f;
                            }
                            float t[BATCH][8];
#pragma unroll
                            for( int s = 0; s < BATCH; ++s )
#pragma unroll
                                for( int c = 0; c < 8; ++c )
                                {
                                    const uint32_t idx = ax[s][c & 1] + ay[s][( c >> 1 ) & 1] + az[s][c >> 2];
                                    t[s][c] = (float)atlas[act[s] ? idx : 0u];
                                }
#pragma unroll
                            for( int s = 0; s < BATCH; ++s )
                            {
                                const float sc = 1.0f / 16777216.0f;
                                const float wx = (float)( wfx[s] & 0xFFFFFFu ) * sc;
                                const float wy = (float)( wfy[s] & 0xFFFFFFu ) * sc;
                                const float wz = (float)( wfz[s] & 0xFFFFFFu ) * sc;
                                e[s] = vrc_classify( lut, vrc_trilerp( t[s], wx, wy, wz ), cls );
                            }
                        }
                        else
                        {
                            uint32_t idx[BATCH], d[BATCH];
#pragma unroll
                            for( int s = 0; s < BATCH; ++s )
                            {
                                const bool take = strag && ( b0 + s < g );
                                act[s] = take && travel > 0.0f;
                                const uint32_t iv = vrc_voxel_address( sm, fx >> 24, fy >> 24, fz >> 24 );
                                idx[s] = act[s] ? iv : 0u;
                                fx += take ? fdx : 0u;
                                fy += take ? fdy : 0u;
                                fz += take ? fdz : 0u;
                                travel -= take ? stepSize : 0.0f;
                            }
#pragma unroll
                            for( int s = 0; s < BATCH; ++s )
                                d[s] = (uint32_t)atlas[idx[s]];
#pragma unroll
                            for( int s = 0; s < BATCH; ++s )
                                e[s] = lut[act[s] ? d[s] : 256u];
                        }
#pragma unroll
                        for( int s = 0; s < BATCH; ++s )
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
            if( hasSeg && ( done || !( travel > 0.0f ) ) )
                hasSeg = false;
            __builtin_amdgcn_wave_barrier(); /* the next round overwrites the region */
        }
    }

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

hipError_t vrc_launch_raycast_lds( const vrc_raycast_args& a, hipStream_t stream )
{
    const uint32_t tilesX = ( a.frame.width + 7u ) / 8u;
    const uint32_t tilesY = ( a.frame.height + 7u ) / 8u;
    const uint32_t nTiles = tilesX * tilesY;
    if( nTiles == 0 )
        return hipSuccess;
    const dim3 grid( ( vrc_schedule_slots( tilesX, tilesY ) + VRC_LDS_WAVES - 1u ) / VRC_LDS_WAVES ),
        block( 64u * VRC_LDS_WAVES );
    const bool count = a.sampleCounter != nullptr;
    /* region shape from the view direction in volume space (the ray through the frame centre) */
    const vrc_ray centre = vrc_setup_ray( a.frame, a.frame.width / 2u, (uint32_t)( a.frame.vpH * 0.5f ) );
    bool flat = fabsf( centre.dir.y ) <= 0.94f;
#if defined( VRC_DEV_KNOBS ) /* A/B builds only (tools/build_variants.sh): VRC_LDS_SHAPE=flat|deep */
    if( const char* force = getenv( "VRC_LDS_SHAPE" ) )
        flat = force[0] == 'f';
#endif
#define VRC_LDS_LAUNCH( COUNT, LINEAR, SHAPE, GREY )                                                         \
    hipLaunchKernelGGL( ( vrc_k_raycast_lds< COUNT, LINEAR, SHAPE, GREY > ), grid, block, 0, stream, a.frame, \
                        a.nodes, a.gridTable, (const uint8_t*)a.atlas, a.lut, a.classifier,                  \
                        a.pixelBuffer, a.sampleCounter, a.tileOrder, tilesX, nTiles )
#define VRC_LDS_LAUNCH_SHAPE( COUNT, LINEAR )                                         \
    {                                                                                 \
        if( a.greyTable )                                                             \
        {                                                                             \
            if( flat ) VRC_LDS_LAUNCH( COUNT, LINEAR, vrc_lds_shape_flat, true );       \
            else VRC_LDS_LAUNCH( COUNT, LINEAR, vrc_lds_shape_deep, true );             \
        }                                                                             \
        else if( flat ) VRC_LDS_LAUNCH( COUNT, LINEAR, vrc_lds_shape_flat, false );     \
        else VRC_LDS_LAUNCH( COUNT, LINEAR, vrc_lds_shape_deep, false );                \
    }
    if( a.linear )
    {
        if( count ) VRC_LDS_LAUNCH_SHAPE( true, true ) else VRC_LDS_LAUNCH_SHAPE( false, true )
    }
    else
    {
        if( count ) VRC_LDS_LAUNCH_SHAPE( true, false ) else VRC_LDS_LAUNCH_SHAPE( false, false )
    }
#undef VRC_LDS_LAUNCH_SHAPE
#undef VRC_LDS_LAUNCH
    return hipGetLastError();
}

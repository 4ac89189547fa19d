/*
 * vrc_core.h -- per-ray arithmetic of the raycaster, shared by the gfx950 kernels
 * (vrc_kernels.hip) and by a host-compiled unit harness (tests/cpu_harness.cpp, g++,
 * used only to exercise this logic under sanitizers where no GPU exists; it is never
 * loaded by the product).
 *
 * What the arithmetic must reproduce is the reference kernel
 * renderers/cudaRaycaster/cuda/Renderer.cu:95-230 (cited per function).  How it is
 * organised is new: a brick-grid DDA instead of the O(nodes) loop, a per-frame 256-entry
 * "classified sample" table instead of a per-sample TF fetch + pow, and an atlas stored as
 * 8x8x8-voxel micro-blocks.
 */
#ifndef VRC_CORE_H
#define VRC_CORE_H

/* ---- one guard for every compile-time switch of the kernels (round 3) ------------------------------------------
 * The product is built with NONE of them on the command line (__graft_entry__.build()).  Timing ablations that
 * render wrong pixels on purpose (VRC_ABLATE_*), alternative layouts and schedules, statistics builds, the biased
 * negative control of the parity tests and every tuning value live behind VRC_DEV_BUILD: a stray -DVRC_... without
 * it does not compile, and a library built with it says so (vrc_abi_version() < 0, vrc_is_dev_build() = 1;
 * bench.py refuses it).  tools/build_variants.sh and the test harnesses pass -DVRC_DEV_BUILD. */
#if !defined( VRC_DEV_BUILD ) && ( \
    defined( VRC_ABLATE_NO_FETCH ) || \
    defined( VRC_ABLATE_HALF_FETCH ) || \
    defined( VRC_ABLATE_QUARTER_FETCH ) || \
    defined( VRC_ZRUN ) || \
    defined( VRC_SETPRIO ) || \
    defined( VRC_INT_STEPS ) || defined( VRC_WG_TIMELINE ) || \
    defined( VRC_PIPELINE ) || \
    defined( VRC_LAYOUT ) || \
    defined( VRC_LAYOUT_PADX ) || \
    defined( VRC_LANES_ROWMAJOR ) || \
    defined( VRC_NO_SERIAL_STEPS ) || \
    defined( VRC_NO_ADDR_TABLES ) || \
    defined( VRC_ADDR_TABLES ) || \
    defined( VRC_TEST_BIAS_ENTRY ) || \
    defined( VRC_DEV_KNOBS ) || \
    defined( VRC_LDS_STATS ) || defined( VRC_LDS_STATS2 ) || \
    defined( VRC_LDS_TIMING ) || \
    defined( VRC_LDS_ABLATE_GATHER ) || \
    defined( VRC_LDS_NO_STEP_CAP ) || defined( VRC_LDS_LOAD_ALL ) || \
    defined( VRC_LDS_PASSES ) || defined( VRC_LDS_ROWS ) || defined( VRC_LDS_OCC ) || defined( VRC_LDS_STAGE_N ) || \
    defined( VRC_LDS_ROWS16 ) || defined( VRC_LDS_OCC16 ) || defined( VRC_LDS_STAGE_N16 ) || \
    defined( VRC_LDS_KMAX ) || \
    defined( VRC_LDS_MAX_DZ ) || \
    defined( VRC_LDS_LBATCH ) || \
    defined( VRC_LDS_GBATCH ) || \
    defined( VRC_LDS_G ) || defined( VRC_LDS_GF ) || \
    defined( VRC_LDS_WAVES ) || \
    defined( VRC_LDS_REFILL ) || defined( VRC_LDS_SOON ) || \
    defined( VRC_GROUP ) || \
    defined( VRC_GREY_GROUP ) || \
    defined( VRC_GREY_MAX_WAVES ) || \
    defined( VRC_TAIL_GROUP ) || \
    defined( VRC_LGROUP ) || \
    defined( VRC_PGROUP ) || defined( VRC_PGROUP16 ) || defined( VRC_PACKED_WAVES ) || defined( VRC_PACKED_ABLATE ) || \
    defined( VRC_SPLIT_GROUP ) || \
    defined( VRC_SMALL_GROUP ) || \
    defined( VRC_SMALL_LAUNCH_TILES ) || \
    defined( VRC_TILE_W ) || defined( VRC_SUPER_UNITS ) || \
    defined( VRC_WAVES_PER_WG ) || \
    defined( VRC_MIN_WAVES ) || \
    defined( VRC_RL_WAVES ) || \
    defined( VRC_RL_MIN_BLOCKS ) )
#error "developer switches of the raycast kernels need -DVRC_DEV_BUILD (the product is built without any of them)"
#endif


#include <stdint.h>

#if defined( __HIPCC__ )
#define VRC_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#include <string.h>
#define VRC_HD inline
#endif

/* Ray and brick-segment set-up is evaluated without FMA contraction, operation for operation
 * as the oracle does: the first sample of every brick segment lies exactly on a brick face
 * (= a voxel face), so one ulp in rayStart decides which of two voxels it reads.  The
 * per-sample loop keeps contraction (it is the hot path). */
#if defined( __clang__ )
#pragma clang fp contract( off ) /* file scope: everything below unless re-enabled */
#define VRC_STRICT_FP _Pragma( "clang fp contract(off)" )
#define VRC_FAST_FP _Pragma( "clang fp contract(fast)" )
#else
#define VRC_STRICT_FP
#define VRC_FAST_FP
#endif

#define VRC_VARIANT_CUDA 0u
#define VRC_VARIANT_GL 1u
#define VRC_EARLY_EXIT 0.999f     /* Renderer.cu:34 */
#define VRC_EPSILON 0.0000000001f /* Renderer.cu:35 */

/* per-ray LOD: levels of the brick hierarchy; entries of one classified table */
#define VRC_MAX_LOD_LEVELS 8
#define VRC_LUT_ENTRIES 257u

/* atlas micro-block: 8x8x8 voxels, x fastest inside the block */
#define VRC_MB 8u
#define VRC_MB_SHIFT 3u
#define VRC_MB_VOXELS 512u

struct vrc_f3
{
    float x, y, z;
};
struct vrc_f4
{
    float x, y, z, w;
};
/* (grey, alpha): a colour or a classified-table entry whose red, green and blue are one number (VRC_MODE_GREY) */
struct vrc_f2
{
    float x, w;
};

/* Kernel-side copy of vrc_node_data plus what the sampler needs per brick. 64 bytes. */
struct vrc_dev_node
{
    float aabbMin[3];
    float aabbSize[3];
    float voxPerWorld[3]; /* texSize*atlasDim/aabbSize: atlas voxels per world unit */
    float localOrigin[3]; /* atlas-voxel coordinate of aabbMin, relative to the slot origin */
    uint32_t slotBase;    /* element offset of the brick's slot in the atlas buffer, low 32 bits */
    uint32_t level;       /* per-ray LOD: 0 = the finest voxel size in the node list, +1 per doubling */
    uint32_t slotBaseHi;  /* high 32 bits: non-zero only in atlases of more than 2^32 voxels (BIG kernels) */
    uint32_t pad;
};

/* Frame constants, derived on the host exactly as Renderer.cu:159-170 does per thread. */
struct vrc_frame
{
    float eye[3];
    float vpX, vpY, vpW, vpH;      /* glViewport as floats */
    float pixelOffX, pixelOffY;    /* sort-first tile offset added to the buffer-local pixel */
    float invProj[16];
    float invView[16];
    float aabbMin[3], aabbMax[3];
    float nearPlane;
    float stepSize;
    uint32_t width, height;        /* pixel buffer */
    uint32_t nPlanes;
    float planes[6][4];
    uint32_t nodeCount;
    /* atlas */
    uint32_t sbx, sby;             /* micro-blocks per slot row / column (slotDim/8) */
    uint32_t slotDim[3];           /* padded slot size in voxels */
    /* brick grid for the DDA kernel */
    float gridMin[3];
    float cellSize[3];
    float invCellSize[3];
    int32_t gridDim[3];
    /* sort-first row bands: frame row of every row of the pixel buffer (NULL = identity).  The
     * buffer then holds height rows picked from a frame of vpH rows (vrc_set_row_map). */
    const uint32_t* rowMap;
    /* 0: cudaRaycaster semantics (cuda/Renderer.cu); 1: the GLSL twin's
     * (glRaycaster/shaders/fragRaycast.glsl), see vrc_brick_segment */
    uint32_t variant;
    /* glRaycaster variant only: nSamplesPerPixel of fragRaycast.glsl:121-129, :212-214 (>= 1).  More than one:
     * every brick is marched by that many jittered rays per pixel and the pixel becomes their average, brick by
     * brick (vrc_pixel_gl_supersampled); the CUDA kernel ignores RenderData.samplesPerPixel (quirk Q3) */
    uint32_t samplesPerPixel;
    /* 1: first march into a pixel buffer that vrc_pre_render declared cleared but did not touch:
     * every pixel starts from 0 and is stored, hit or miss (the clear of
     * cuda/PixelBufferObject.cu:80 folded into the march: one pass over the frame less) */
    uint32_t clearFirst;
    /* per-ray LOD (vrc_pixel_ray_lod): number of levels in the node list,
     * finestVoxelWorldSize / (screenSpaceError * worldSpacePerPixel), and one brick grid per
     * level anchored at gridMin (gridDim / cellSize are level 0's): cells per axis, 1 / cell size,
     * offset of the level's cell -> node table in the table buffer; lodMax = max corner of all
     * bricks; lodEps = 1 % of the finest voxel */
    uint32_t lodLevels;
    float lodBase;
    float lodEps;
    float lodMax[3];
    float lodInvCell[VRC_MAX_LOD_LEVELS][3];
    int32_t lodDim[VRC_MAX_LOD_LEVELS][3];
    uint32_t lodTable[VRC_MAX_LOD_LEVELS];
};

/* Atlas memory layout.  The logical atlas is the reference's 3-D array of slots
 * (cuda/TexturePool.cu:128-150); physically every slot is one contiguous run of
 * slotDim.x*slotDim.y*slotDim.z elements, and inside a slot voxels are grouped in 8x8x8
 * micro-blocks (x fastest inside a block, blocks x-fastest inside the slot).  A brick is
 * therefore one contiguous 2.4 MiB range (TLB-friendly) and the 64 fetches of a wave step
 * land in a few 64-byte segments. */
struct vrc_layout
{
    uint32_t slots[3];   /* slot grid */
    uint32_t slotDim[3]; /* padded slot size in voxels, multiples of 8 */
};

/* ---- slot-local layout: element offset = LX(x) + LY(y) + LZ(z), one part per axis ---------------
 * VRC_LAYOUT 0 (the product's layout): 8x8x8 micro-blocks, blocks x-fastest inside the slot; inside a block the
 * 64 x-rows of 8 voxels lie in the order y & 3, z & 3, y >> 2, z >> 2, so that a 128-byte line -- what an L2 fetches
 * from HBM -- is a block of 8 x 4 x 4 voxels and the lines a tile touches per step are the same for views along every
 * axis (round 4; rounds 1-3 had the rows in the order y, z -- lines of 8 x 8 x 2 -- and ran views along x and y
 * 17 % / 7 % slower than along z: profiles/r4_byte_atlas_line_shapes.txt).  Rows y and y + 1 (y even) stay neighbours.
 * Other values are developer experiments (tools/dev_layouts.sh): only the table-driven point-sampling
 * kernel understands them.
 *   1: micro-blocks 576 B apart and block rows padded to 3 mod 8 blocks: the 64-B lines of x/y
 *      neighbour blocks fall into different 64-B phases of a 512-B period
 *   2: row-major (x fastest), pitch slotDim.x + VRC_LAYOUT_PADX
 *   3: 64-B lines of 32 x 2 voxels      4: 64-B lines of 16 x 4 voxels
 *   5: 64-B lines of 4 x 4 x 4 voxels, z fastest (a dword = four voxels along z); with VRC_ZRUN the fast groups of
 *      the march keep a lane's dword and reload it only when the voxel column or the 4-voxel block changes
 *   6: micro-blocks with the rows in the order y, z (rounds 1-3: lines of 8 x 8 x 2)
 *   7: ... in the order y & 1, z, y >> 1 (lines of 8 x 2 x 8) */
/* in-block offset of row y / slice z (bits 0-2 of u count) */
VRC_HD uint32_t vrc_mb_y( uint32_t u ) { return ( ( u & 3u ) << 3 ) | ( ( u & 4u ) << 5 ); }
VRC_HD uint32_t vrc_mb_z( uint32_t u ) { return ( ( u & 3u ) << 5 ) | ( ( u & 4u ) << 6 ); }
/* the same as corrections of the plain order: vrc_mb_y( u ) = 8 (u & 7) + VRC_MB_FIX_Y( u ),
 * vrc_mb_z( u ) = 64 (u & 7) - VRC_MB_FIX_Z( u ) -- what the arithmetic address paths add to 8 y + 64 z */
#define VRC_MB_FIX_Y( u ) ( 96u * ( ( ( u ) >> 2 ) & 1u ) )
#define VRC_MB_FIX_Z( u ) ( 32u * ( ( u ) & 3u ) )
#ifndef VRC_LAYOUT
#define VRC_LAYOUT 0
#endif
#ifndef VRC_LAYOUT_PADX
#define VRC_LAYOUT_PADX 0
#endif
struct vrc_lay
{
    uint32_t a, b, c; /* meaning depends on the layout */
};
VRC_HD vrc_lay vrc_make_lay( uint32_t sbx, uint32_t sby )
{
    vrc_lay l;
#if VRC_LAYOUT == 0 || VRC_LAYOUT == 6
    l.a = 504u; l.b = sbx * VRC_MB_VOXELS - 64u; l.c = sbx * sby * VRC_MB_VOXELS - 512u;
#elif VRC_LAYOUT == 1
    const uint32_t sbxp = sbx + ( ( 3u - sbx ) & 7u );
    l.a = 576u; l.b = 576u * sbxp; l.c = 576u * sbxp * sby;
#elif VRC_LAYOUT == 2
    l.a = 1u; l.b = sbx * 8u + VRC_LAYOUT_PADX; l.c = l.b * sby * 8u;
#elif VRC_LAYOUT == 3
    l.a = 64u; l.b = 64u * ( ( sbx * 8u + 31u ) / 32u ); l.c = l.b * sby * 4u;
#elif VRC_LAYOUT == 5
    l.a = 64u; l.b = 64u * sbx * 2u; l.c = l.b * sby * 2u;
#elif VRC_LAYOUT == 7
    l.a = 504u; l.b = sbx * VRC_MB_VOXELS; l.c = sbx * sby * VRC_MB_VOXELS;
#else
    l.a = 64u; l.b = 64u * ( ( sbx * 8u + 15u ) / 16u ); l.c = l.b * sby * 2u;
#endif
    return l;
}
VRC_HD uint32_t vrc_lay_x( const vrc_lay& l, uint32_t u )
{
#if VRC_LAYOUT == 0 || VRC_LAYOUT == 6 || VRC_LAYOUT == 7
    return u + l.a * ( u >> 3 );
#elif VRC_LAYOUT == 1
    return ( u & 7u ) + l.a * ( u >> 3 );
#elif VRC_LAYOUT == 2
    return u;
#elif VRC_LAYOUT == 3
    return ( u & 31u ) + l.a * ( u >> 5 );
#elif VRC_LAYOUT == 5
    return 16u * ( u & 3u ) + l.a * ( u >> 2 );
#else
    return ( u & 15u ) + l.a * ( u >> 4 );
#endif
}
VRC_HD uint32_t vrc_lay_y( const vrc_lay& l, uint32_t u )
{
#if VRC_LAYOUT == 0
    return 8u * u + l.b * ( u >> 3 ) + VRC_MB_FIX_Y( u );
#elif VRC_LAYOUT == 6
    return 8u * u + l.b * ( u >> 3 );
#elif VRC_LAYOUT == 1
    return 8u * ( u & 7u ) + l.b * ( u >> 3 );
#elif VRC_LAYOUT == 2
    return l.b * u;
#elif VRC_LAYOUT == 3
    return 32u * ( u & 1u ) + l.b * ( u >> 1 );
#elif VRC_LAYOUT == 5
    return 4u * ( u & 3u ) + l.b * ( u >> 2 );
#elif VRC_LAYOUT == 7
    return 8u * ( ( u & 1u ) | ( ( ( u >> 1 ) & 3u ) << 4 ) ) + l.b * ( u >> 3 );
#else
    return 16u * ( u & 3u ) + l.b * ( u >> 2 );
#endif
}
VRC_HD uint32_t vrc_lay_z( const vrc_lay& l, uint32_t u )
{
#if VRC_LAYOUT == 0
    return 64u * u + l.c * ( u >> 3 ) - VRC_MB_FIX_Z( u );
#elif VRC_LAYOUT == 6
    return 64u * u + l.c * ( u >> 3 );
#elif VRC_LAYOUT == 1
    return 64u * ( u & 7u ) + l.c * ( u >> 3 );
#elif VRC_LAYOUT == 5
    return ( u & 3u ) + l.c * ( u >> 2 );
#elif VRC_LAYOUT == 7
    return 8u * ( ( u & 7u ) << 1 ) + l.c * ( u >> 3 );
#else
    return l.c * u;
#endif
}
/* physical elements of one slot */
VRC_HD uint64_t vrc_slot_elems( uint32_t sdx, uint32_t sdy, uint32_t sdz )
{
#if VRC_LAYOUT == 0 || VRC_LAYOUT == 6 || VRC_LAYOUT == 7
    return (uint64_t)sdx * sdy * sdz;
#elif VRC_LAYOUT == 1
    return (uint64_t)vrc_make_lay( sdx >> 3, sdy >> 3 ).c * ( sdz >> 3 );
#elif VRC_LAYOUT == 5
    return (uint64_t)vrc_make_lay( sdx >> 3, sdy >> 3 ).c * ( sdz >> 2 );
#else
    return (uint64_t)vrc_make_lay( sdx >> 3, sdy >> 3 ).c * sdz;
#endif
}

/* element offset of voxel (x,y,z), local to a slot of sbx x sby x * micro-blocks */
VRC_HD uint32_t vrc_slot_local_index( uint32_t x, uint32_t y, uint32_t z, uint32_t sbx, uint32_t sby )
{
#if VRC_LAYOUT == 0
    const uint32_t blk = ( ( z >> VRC_MB_SHIFT ) * sby + ( y >> VRC_MB_SHIFT ) ) * sbx +
                         ( x >> VRC_MB_SHIFT );
    const uint32_t inner = vrc_mb_z( z ) | vrc_mb_y( y ) | ( x & 7u );
    return blk * VRC_MB_VOXELS + inner;
#else
    const vrc_lay l = vrc_make_lay( sbx, sby );
    return vrc_lay_x( l, x ) + vrc_lay_y( l, y ) + vrc_lay_z( l, z );
#endif
}

/* element offset of slot (i,j,k): 64 bits, an atlas may hold more than 2^32 voxels (it is sized
 * for the GPU's memory, not for a 32-bit index; the reference truncates, quirk Q12) */
VRC_HD uint64_t vrc_slot_base( const vrc_layout& l, uint32_t i, uint32_t j, uint32_t k )
{
    const uint64_t slotVoxels = vrc_slot_elems( l.slotDim[0], l.slotDim[1], l.slotDim[2] );
    return ( ( (uint64_t)k * l.slots[1] + j ) * l.slots[0] + i ) * slotVoxels;
}

/* physical element index of logical atlas voxel (x,y,z) */
VRC_HD uint64_t vrc_atlas_index( const vrc_layout& l, uint32_t x, uint32_t y, uint32_t z )
{
    const uint32_t i = x / l.slotDim[0], j = y / l.slotDim[1], k = z / l.slotDim[2];
    return vrc_slot_base( l, i, j, k ) +
           vrc_slot_local_index( x - i * l.slotDim[0], y - j * l.slotDim[1], z - k * l.slotDim[2],
                                 l.slotDim[0] >> VRC_MB_SHIFT, l.slotDim[1] >> VRC_MB_SHIFT );
}

/* ------------------------------------------------------------------------------------------
 * Classified-sample table entry for u8 density d: what Renderer.cu:215-218 computes from a
 * sample of that density: tf = tex1D(tfTex, d*mult+add) (cuda/ColorMap.cu:40-45: 256 texels,
 * linear, normalized, clamp; the lerp weight is kept in fracBits fractional bits, 8 on CUDA
 * hardware, 0 = exact float); alpha' = 1 - pow(1 - min(tf.a, 255/256), alphaCorrection)
 * (Renderer.cu:88-89); entry = (tf.rgb*alpha', alpha').  A pure function of d within a frame.
 * ---------------------------------------------------------------------------------------- */
struct vrc_lut_params
{
    float rangeMin, rangeMax; /* RenderData.dataSourceRange, Renderer.cu:162-164 */
    float alphaCorrection;    /* maxSamplesPerRay / samplesPerRay, Renderer.cu:167-168 */
    int fracBits;
};

VRC_HD vrc_f4 vrc_lut_entry( const float* tf, uint32_t d, vrc_lut_params p )
{
#if defined( __clang__ )
#pragma clang fp contract( off )
#endif
    const float multiplyer = 1.0f / ( p.rangeMax - p.rangeMin );
    const float addedValue = -p.rangeMin / ( p.rangeMax - p.rangeMin );
    const float u = (float)d * multiplyer + addedValue;
    const float xB = u * 256.0f - 0.5f;
    const float fl = floorf( xB );
    float a = xB - fl;
    if( p.fracBits > 0 )
    {
        const float q = (float)( 1 << p.fracBits );
        a = floorf( a * q + 0.5f ) / q;
    }
    int i0 = (int)fl, i1 = (int)fl + 1;
    i0 = i0 < 0 ? 0 : ( i0 > 255 ? 255 : i0 );
    i1 = i1 < 0 ? 0 : ( i1 > 255 ? 255 : i1 );
    float c[4];
    for( int k = 0; k < 4; ++k )
        c[k] = ( 1.0f - a ) * tf[i0 * 4 + k] + a * tf[i1 * 4 + k];
    const float corr = 1.0f - fminf( c[3], 1.0f - 1.0f / 256.0f );
    const float alpha = 1.0f - powf( corr, p.alphaCorrection );
    vrc_f4 e;
    e.x = c[0] * alpha;
    e.y = c[1] * alpha;
    e.z = c[2] * alpha;
    e.w = alpha;
    return e;
}

/* cuda/math.cuh:1457-1464 (column-major 4x4 times vec4) */
VRC_HD vrc_f4 vrc_mul44( const float* m, vrc_f4 v )
{
    VRC_STRICT_FP
    vrc_f4 r;
    r.x = m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * v.w;
    r.y = m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * v.w;
    r.z = m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * v.w;
    r.w = m[3] * v.x + m[7] * v.y + m[11] * v.z + m[15] * v.w;
    return r;
}

VRC_HD float vrc_dot( vrc_f3 a, vrc_f3 b )
{
    VRC_STRICT_FP
    return a.x * b.x + a.y * b.y + a.z * b.z;
}

/* cuda/math.cuh:1310-1314 */
VRC_HD vrc_f3 vrc_normalize( vrc_f3 v )
{
    VRC_STRICT_FP
    const float invLen = 1.0f / sqrtf( vrc_dot( v, v ) );
    vrc_f3 r = { v.x * invLen, v.y * invLen, v.z * invLen };
    return r;
}

/* Renderer.cu:56-80; invR is hoisted (the reference recomputes the same value per call) */
VRC_HD bool vrc_intersect_box( vrc_f3 origin, vrc_f3 invR, vrc_f3 boxMin, vrc_f3 boxMax,
                               float* tnear, float* tfar )
{
    VRC_STRICT_FP
    const float tbx = invR.x * ( boxMin.x - origin.x ), ttx = invR.x * ( boxMax.x - origin.x );
    const float tby = invR.y * ( boxMin.y - origin.y ), tty = invR.y * ( boxMax.y - origin.y );
    const float tbz = invR.z * ( boxMin.z - origin.z ), ttz = invR.z * ( boxMax.z - origin.z );
    const float tminx = fminf( ttx, tbx ), tmaxx = fmaxf( ttx, tbx );
    const float tminy = fminf( tty, tby ), tmaxy = fmaxf( tty, tby );
    const float tminz = fminf( ttz, tbz ), tmaxz = fmaxf( ttz, tbz );
    const float largestTmin = fmaxf( fmaxf( tminx, tminy ), tminz );
    const float smallestTmax = fminf( fminf( tmaxx, tmaxy ), tmaxz );
    *tnear = largestTmin;
    *tfar = smallestTmax;
    return smallestTmax > largestTmin;
}

struct vrc_ray
{
    vrc_f3 origin, dir, invDir;
    float tNearGlobal, tFarGlobal, tNearPlane;
    bool hit;
};

/* Renderer.cu:106-149 + :159-160: pixel -> world ray, global box, clip planes, near plane */
/* wx, wy: window-space position the ray is cast through (gl_FragCoord of the GLSL twin; its jittered
 * sub-pixel positions, fragRaycast.glsl:123-127) */
VRC_HD vrc_ray vrc_setup_ray_at( const vrc_frame& f, float wx, float wy )
{
    VRC_STRICT_FP
    vrc_ray r;
    /* Renderer.cu:40-51 */
    const float nx = 2.0f * ( wx - f.vpX - ( f.vpW / 2.0f ) ) / f.vpW;
    const float ny = 2.0f * ( wy - f.vpY - ( f.vpH / 2.0f ) ) / f.vpH;
    const vrc_f4 ndc = { nx, ny, 1.0f, 1.0f };
    const vrc_f4 e = vrc_mul44( f.invProj, ndc );
    const vrc_f4 eyeSpace = { e.x / e.w, e.y / e.w, e.z / e.w, e.w / e.w };
    const vrc_f4 world = vrc_mul44( f.invView, eyeSpace );
    r.origin.x = f.eye[0];
    r.origin.y = f.eye[1];
    r.origin.z = f.eye[2];
    const vrc_f3 d0 = { world.x - r.origin.x, world.y - r.origin.y, world.z - r.origin.z };
    r.dir = vrc_normalize( d0 );
    if( r.dir.x == 0.0f ) r.dir.x = VRC_EPSILON;
    if( r.dir.y == 0.0f ) r.dir.y = VRC_EPSILON;
    if( r.dir.z == 0.0f ) r.dir.z = VRC_EPSILON;
    r.invDir.x = 1.0f / r.dir.x;
    r.invDir.y = 1.0f / r.dir.y;
    r.invDir.z = 1.0f / r.dir.z;

    const vrc_f3 gmin = { f.aabbMin[0], f.aabbMin[1], f.aabbMin[2] };
    const vrc_f3 gmax = { f.aabbMax[0], f.aabbMax[1], f.aabbMax[2] };
    r.hit = vrc_intersect_box( r.origin, r.invDir, gmin, gmax, &r.tNearGlobal, &r.tFarGlobal );
    if( f.variant == VRC_VARIANT_GL ) /* fragRaycast.glsl:101: t0 <= t1 */
        r.hit = r.tNearGlobal <= r.tFarGlobal;

    /* Renderer.cu:132-146; the GLSL twin clips per brick instead (fragRaycast.glsl:162-174) */
    for( uint32_t i = 0; i < ( f.variant == VRC_VARIANT_GL ? 0u : f.nPlanes ); ++i )
    {
        const vrc_f3 n = { f.planes[i][0], f.planes[i][1], f.planes[i][2] };
        float rn = vrc_dot( r.dir, n );
        if( rn == 0.0f )
            rn = VRC_EPSILON;
        const float t = -( vrc_dot( n, r.origin ) + f.planes[i][3] ) / rn;
        if( rn > 0.0f )
            r.tNearGlobal = fmaxf( r.tNearGlobal, t );
        else
            r.tFarGlobal = fminf( r.tFarGlobal, t );
    }
    if( r.tNearGlobal > r.tFarGlobal )
        r.hit = false;

    /* Renderer.cu:159-160 */
    const vrc_f3 e3 = { eyeSpace.x, eyeSpace.y, eyeSpace.z };
    const vrc_f3 ne = vrc_normalize( e3 );
    r.tNearPlane = -f.nearPlane / ne.z;
    return r;
}

VRC_HD vrc_ray vrc_setup_ray( const vrc_frame& f, uint32_t px, uint32_t py )
{
    VRC_STRICT_FP
    return vrc_setup_ray_at( f, (float)px + f.pixelOffX, (float)py + f.pixelOffY );
}

/* rand() of fragRaycast.glsl:59-62: fract(sin(dot(co, vec2(12.9898, 78.233))) * 43758.5453).  What a GL
 * implementation's sin() returns for arguments of 1e4..1e6 is implementation-defined to more bits than the
 * factor 43758 leaves, so no two GPUs jitter alike; THIS build defines it with the sine evaluated in double
 * (libm and the device library agree to the float), the rest in float without contraction -- the oracle's
 * gl_rand, bit for bit. */
VRC_HD float vrc_gl_rand( float x, float y )
{
    VRC_STRICT_FP
    const float d = x * 12.9898f + y * 78.233f;
    const float sn = (float)sin( (double)d );
    const float v = sn * 43758.5453f;
    return v - floorf( v );
}

/* One brick segment of one ray, Renderer.cu:179-201: returns false if the brick is skipped.
 * stop: set when the reference would leave the node loop (tNear > tFarGlobal). */
struct vrc_segment
{
    vrc_f3 pos;   /* rayStart */
    vrc_f3 step;  /* normalize(stop-start)*stepSize */
    float dist;
    float tNear;  /* ray parameter of rayStart */
};

VRC_HD bool vrc_brick_segment( const vrc_frame& f, const vrc_ray& r, const vrc_dev_node& n,
                               float stepSize, vrc_segment* s, bool* stop )
{
    VRC_STRICT_FP
    const vrc_f3 boxMin = { n.aabbMin[0], n.aabbMin[1], n.aabbMin[2] };
    const vrc_f3 boxMax = { boxMin.x + n.aabbSize[0], boxMin.y + n.aabbSize[1],
                            boxMin.z + n.aabbSize[2] };
    float tNear = 0.0f, tFar = 0.0f;
    *stop = false;
    if( f.variant == VRC_VARIANT_GL )
    {
        /* fragRaycast.glsl:142-177: hit test t0 <= t1; tnear raised to the near plane only;
         * first sample snapped to the lattice tnearGlobal + k*stepSize; clip planes move this
         * brick's interval, after the snap */
        (void)vrc_intersect_box( r.origin, r.invDir, boxMin, boxMax, &tNear, &tFar );
        if( !( tNear <= tFar ) )
            return false;
        if( tNear < r.tNearPlane )
            tNear = r.tNearPlane;
        const float a = tNear - r.tNearGlobal;
        const float residu = a - stepSize * floorf( a / stepSize );
        if( residu > 0.0f )
            tNear += stepSize - residu;
        if( tNear > tFar )
            return false;
        for( uint32_t i = 0; i < f.nPlanes; ++i )
        {
            const vrc_f3 pn = { f.planes[i][0], f.planes[i][1], f.planes[i][2] };
            float rn = vrc_dot( r.dir, pn );
            if( rn == 0.0f )
                rn = VRC_EPSILON;
            const float t = -( vrc_dot( pn, r.origin ) + f.planes[i][3] ) / rn;
            if( rn > 0.0f )
                tNear = fmaxf( tNear, t );
            else
                tFar = fminf( tFar, t );
        }
        if( tNear > tFar )
            return false;
    }
    else
    {
        if( !vrc_intersect_box( r.origin, r.invDir, boxMin, boxMax, &tNear, &tFar ) )
            return false;
        if( tNear > r.tFarGlobal )
        {
            *stop = true;
            return false;
        }
        if( tFar < r.tNearGlobal )
            return false;
        tNear = fmaxf( fmaxf( r.tNearPlane, tNear ), r.tNearGlobal );
        tFar = fminf( tFar, r.tFarGlobal );
        if( tNear > tFar )
            return false;
    }

    const vrc_f3 rayStart = { r.origin.x + r.dir.x * tNear, r.origin.y + r.dir.y * tNear,
                              r.origin.z + r.dir.z * tNear };
    const vrc_f3 rayStop = { r.origin.x + r.dir.x * tFar, r.origin.y + r.dir.y * tFar,
                             r.origin.z + r.dir.z * tFar };
    const vrc_f3 diff = { rayStop.x - rayStart.x, rayStop.y - rayStart.y,
                          rayStop.z - rayStart.z };
    const float d2 = vrc_dot( diff, diff );
    const float invLen = 1.0f / sqrtf( d2 );
    s->pos = rayStart;
#if defined( VRC_DEV_BUILD ) && defined( VRC_TEST_BIAS_ENTRY )
    /* NEGATIVE CONTROL of the parity rule (tests/test_cpu_harness.py::test_parity_rule_rejects_a_biased_kernel; never in
     * the product: __graft_entry__.build() does not define VRC_DEV_BUILD): every brick segment starts 2e-7 world units
     * early, so its first sample -- the reference puts it exactly on the brick face, cuda/Renderer.cu:195-196 -- always
     * reads the voxel on the near side of the face.  Every such flip is inside the oracle's tie zone; the rule must
     * still reject a kernel that takes all of them. */
    s->pos.x -= r.dir.x * 2e-7f;
    s->pos.y -= r.dir.y * 2e-7f;
    s->pos.z -= r.dir.z * 2e-7f;
#endif
    s->step.x = diff.x * invLen * stepSize;
    s->step.y = diff.y * invLen * stepSize;
    s->step.z = diff.z * invLen * stepSize;
    s->dist = sqrtf( d2 );
    s->tNear = tNear;
    return true;
}

/* 24-bit multiply (v_mul_u32_u24 / v_mad_u32_u24 on gfx950: full rate, where a 32-bit
 * multiply is quarter rate). */
VRC_HD uint32_t vrc_mul24( uint32_t a, uint32_t b )
{
#if defined( __HIP_DEVICE_COMPILE__ )
    return __umul24( a, b );
#else
    return ( a & 0xFFFFFFu ) * ( b & 0xFFFFFFu );
#endif
}

/* Per-brick sampler constants, hoisted out of the march loop. */
struct vrc_sampler
{
    float minx, miny, minz;   /* aabbMin */
    float kx, ky, kz;         /* atlas voxels per world unit */
    float ox, oy, oz;         /* slot-local voxel coordinate of aabbMin (= overlap) */
    float hix, hiy, hiz;      /* slotDim - 1 (clamped sampler only) */
    uint32_t slotBase;        /* element offset of the slot */
    uint32_t cyy, czz;        /* micro-block stride in y / z minus the in-block stride * 8 */
};

VRC_HD vrc_sampler vrc_make_sampler( const vrc_dev_node& n, const vrc_frame& f )
{
    vrc_sampler s;
    s.minx = n.aabbMin[0]; s.miny = n.aabbMin[1]; s.minz = n.aabbMin[2];
    s.kx = n.voxPerWorld[0]; s.ky = n.voxPerWorld[1]; s.kz = n.voxPerWorld[2];
    s.ox = n.localOrigin[0]; s.oy = n.localOrigin[1]; s.oz = n.localOrigin[2];
    s.hix = (float)( f.slotDim[0] - 1u );
    s.hiy = (float)( f.slotDim[1] - 1u );
    s.hiz = (float)( f.slotDim[2] - 1u );
    s.slotBase = n.slotBase;
    s.cyy = f.sbx * VRC_MB_VOXELS - 64u;
    s.czz = f.sbx * f.sby * VRC_MB_VOXELS - 512u;
    return s;
}

/* brick-local voxel of a world position (nearest, Renderer.cu:210-214 + point sampling of
 * cuda/TexturePool.cu:163-170), as the physical element index in the atlas.
 * Evaluated brick-locally so the float grid is finer than the reference's normalized
 * atlas coordinate; voxel choice can differ only for samples within ~1e-4 voxel of a
 * voxel face ("nearest-voxel flips").  Coordinates are >= 0 here, so the
 * float->int truncation is the floor of the point-sampling rule. */
/* element offset of slot-local voxel (ux,uy,uz) */
VRC_HD uint32_t vrc_voxel_address( const vrc_sampler& s, uint32_t ux, uint32_t uy, uint32_t uz )
{
    /* element = slotBase + sum over axes of (c >> 3) * blockStride + (c & 7) * inStride with
     * inStride = 1, 8, 64 and blockStride = 512, 512*sbx, 512*sbx*sby.  Since
     * (c & 7) * s = c * s - (c >> 3) * 8 * s this is
     *   slotBase + x + 8 y + 64 z + qx * 504 + qy * (512 sbx - 64) + qz * (512 sbx sby - 512):
     * three shifts, three 24-bit multiply-adds (full rate; operands < 2^24: slot-local block
     * counts <= 512, slot < 16 Mi elements, checked at pool creation), two shift-adds, one add --
     * plus the row order's corrections (VRC_MB_FIX_Y / _Z: the rows of a block lie y & 3, z & 3, y >> 2, z >> 2). */
#if defined( __HIP_DEVICE_COMPILE__ )
    /* spelled out: hipcc otherwise emits multiply + 3-input add pairs (eleven instructions).
     * v_mad_u32_u24 takes one scalar operand (the stride). */
    uint32_t e, t;
    asm( "v_mad_u32_u24 %0, %1, %2, %3" : "=v"( e ) : "v"( ux >> VRC_MB_SHIFT ), "s"( 504u ), "v"( ux ) );
    asm( "v_mad_u32_u24 %0, %1, %2, %3" : "=v"( t ) : "v"( uy >> VRC_MB_SHIFT ), "s"( s.cyy ), "v"( e ) );
    asm( "v_mad_u32_u24 %0, %1, %2, %3" : "=v"( e ) : "v"( uz >> VRC_MB_SHIFT ), "s"( s.czz ), "v"( t ) );
    asm( "v_lshl_add_u32 %0, %1, 3, %2" : "=v"( t ) : "v"( uy ), "v"( e ) );
    asm( "v_lshl_add_u32 %0, %1, 6, %2" : "=v"( e ) : "v"( uz ), "v"( t ) );
    return e + s.slotBase + VRC_MB_FIX_Y( uy ) - VRC_MB_FIX_Z( uz );
#else
    uint32_t e = vrc_mul24( ux >> VRC_MB_SHIFT, 504u ) + ux;
    e = vrc_mul24( uy >> VRC_MB_SHIFT, s.cyy ) + e;
    e = vrc_mul24( uz >> VRC_MB_SHIFT, s.czz ) + e;
    e = ( uy << 3 ) + e;
    e = ( uz << 6 ) + e;
    return e + s.slotBase + VRC_MB_FIX_Y( uy ) - VRC_MB_FIX_Z( uz );
#endif
}

/* Atlas element indices of the next N samples of a segment, advancing pos by N steps.
 * Per sample: brick-local voxel of the world position (nearest, Renderer.cu:210-214 + point
 * sampling of cuda/TexturePool.cu:163-170): l = (pos - aabbMin) * voxelsPerWorld + localOrigin,
 * voxel = trunc(l) (l >= 0, so truncation is the floor of the point-sampling rule), evaluated
 * brick-locally so the float grid is finer than the reference's normalized atlas coordinate;
 * the voxel can differ only for samples within ~1e-4 voxel of a voxel face (DESIGN.md,
 * "nearest-voxel flips").  pos advances by sequential additions of step, as in the reference.
 * Scalar f32 instructions on purpose: on gfx950 a v_pk_add/fma_f32 costs ~2.9x a scalar
 * v_add_f32 (profiles/r1_ubench_valu_issue_costs.txt), so packing two lanes' worth of data
 * into one instruction loses; the file is also built with -fno-slp-vectorize. */
template < bool CLAMP, int N >
VRC_HD void vrc_group_indices( const vrc_sampler& s, vrc_f3& pos, const vrc_f3& step, uint32_t* idx )
{
    VRC_FAST_FP
#pragma unroll
    for( int k = 0; k < N; ++k )
    {
        float lx = ( pos.x - s.minx ) * s.kx + s.ox;
        float ly = ( pos.y - s.miny ) * s.ky + s.oy;
        float lz = ( pos.z - s.minz ) * s.kz + s.oz;
        if( CLAMP )
        {
            lx = fminf( fmaxf( lx, 0.0f ), s.hix );
            ly = fminf( fmaxf( ly, 0.0f ), s.hiy );
            lz = fminf( fmaxf( lz, 0.0f ), s.hiz );
        }
        idx[k] = vrc_voxel_address( s, (uint32_t)(int)lx, (uint32_t)(int)ly, (uint32_t)(int)lz );
        pos.x += step.x;
        pos.y += step.y;
        pos.z += step.z;
    }
}

/* Fixed-point alternative to the float position chain above (VRC_OPT_STEPPING = 1).
 * The slot-local voxel coordinate of the segment's first sample is computed exactly as above
 * (same float expression, so the sample on the brick face picks the same voxel), converted to
 * 8.24 fixed point (exact: l < 256 has at most 24 fraction bits below 2^8), and every further
 * sample adds the per-step voxel increment round(step * voxelsPerWorld * 2^24): one integer
 * add per axis instead of add + subtract + multiply-add + convert.  It is a more accurate
 * evaluation of start + n*step than the reference's float accumulation (error <= n * 2^-25
 * voxel against ~n * 6e-5), but not the same rounding path, so a sample within ~1e-4 voxel
 * of a voxel face may read the neighbouring voxel ("nearest-voxel flips").
 * Not used with the clamped sampler (coordinates must stay in [0, 256)). */
struct vrc_fixpos
{
    uint32_t x, y, z;    /* 8.24 slot-local voxel coordinates */
    uint32_t dx, dy, dz; /* per-step increments (two's complement) */
};

VRC_HD vrc_fixpos vrc_fixpos_init( const vrc_sampler& s, const vrc_f3& pos, const vrc_f3& step )
{
    /* (round 4, VERDICT r3 item 4: evaluated without contraction like the rest of the set-up -- VRC_STRICT_FP -- the
     * tie bias of the C2 noise rows stays 0.212 to the third digit and the kernel time where it was: the bias comes from
     * the truncation below, which at a face entered from above is the near side, not from the multiply-add) */
    VRC_FAST_FP
    vrc_fixpos p;
    const float lx = ( pos.x - s.minx ) * s.kx + s.ox;
    const float ly = ( pos.y - s.miny ) * s.ky + s.oy;
    const float lz = ( pos.z - s.minz ) * s.kz + s.oz;
    p.x = (uint32_t)( lx * 16777216.0f );
    p.y = (uint32_t)( ly * 16777216.0f );
    p.z = (uint32_t)( lz * 16777216.0f );
    p.dx = (uint32_t)(int32_t)rintf( step.x * s.kx * 16777216.0f );
    p.dy = (uint32_t)(int32_t)rintf( step.y * s.ky * 16777216.0f );
    p.dz = (uint32_t)(int32_t)rintf( step.z * s.kz * 16777216.0f );
    return p;
}

#if !defined( VRC_NO_ADDR_TABLES ) && !defined( VRC_ADDR_TABLES )
#define VRC_ADDR_TABLES /* measured 4 % faster on C2 than the arithmetic form */
#endif
#if defined( __HIPCC__ ) && defined( VRC_ADDR_TABLES )
/* Per-axis address parts of vrc_voxel_address as three 256-entry tables in LDS (filled by the
 * raycast kernel): TX[u] = u + 504 (u>>3), TY[u] = 8u + cyy (u>>3) + FIX_Y(u), TZ[u] = 64u + czz (u>>3) - FIX_Z(u).
 * Replaces 3 shifts + 3 24-bit multiply-adds + 2 shift-adds (the slow VALU classes) by 3 masks
 * + 3 LDS reads + one 3-input add per sample. */
__shared__ uint32_t vrc_addr_tab[3 * 256];
#endif

template < int N >
VRC_HD void vrc_group_indices_fixed( const vrc_sampler& s, vrc_fixpos& p, uint32_t* idx )
{
#if defined( __HIP_DEVICE_COMPILE__ ) && defined( VRC_ADDR_TABLES )
    uint32_t tx[N], ty[N], tz[N];
#pragma unroll
    for( int k = 0; k < N; ++k )
    {
        const char* const t = reinterpret_cast< const char* >( vrc_addr_tab );
        tx[k] = *reinterpret_cast< const uint32_t* >( t + ( ( p.x >> 22 ) & 0x3FCu ) );
        ty[k] = *reinterpret_cast< const uint32_t* >( t + 1024 + ( ( p.y >> 22 ) & 0x3FCu ) );
        tz[k] = *reinterpret_cast< const uint32_t* >( t + 2048 + ( ( p.z >> 22 ) & 0x3FCu ) );
        p.x += p.dx;
        p.y += p.dy;
        p.z += p.dz;
#if !defined( VRC_NO_SERIAL_STEPS )
        /* keep the three additions per step: left alone, hipcc rewrites the chain as p + k * d with a
         * v_mul_lo_u32 or a shift-add per step and axis (the slow integer classes, profiles/r1_ubench_valu_issue_costs.txt)
         * to shorten a dependency that the other waves hide anyway */
        asm( "" : "+v"( p.x ), "+v"( p.y ), "+v"( p.z ) );
#endif
    }
#pragma unroll
    for( int k = 0; k < N; ++k )
        idx[k] = tx[k] + ty[k] + tz[k] + s.slotBase;
#else
#pragma unroll
    for( int k = 0; k < N; ++k )
    {
        idx[k] = vrc_voxel_address( s, p.x >> 24, p.y >> 24, p.z >> 24 );
        p.x += p.dx;
        p.y += p.dy;
        p.z += p.dz;
    }
#endif
}

/* Renderer.cu:83-93 with the classified table: e = (rgb*alpha', alpha') for the density.
 * frozen: the ray already crossed the early-exit threshold; the weight is forced to 0 so the
 * step is an exact no-op (x + e*0 == x) without making the table address depend on it. */
VRC_HD void vrc_composite( vrc_f4& c, const vrc_f4& e, bool frozen = false )
{
    VRC_FAST_FP
    float t = 1.0f - c.w;
    t = frozen ? 0.0f : t;
    c.x = c.x + e.x * t;
    c.y = c.y + e.y * t;
    c.z = c.z + e.z * t;
    c.w = c.w + e.w * t;
}
/* the same blend for a grey table: red, green and blue of the four-float form are the same three operations on the
 * same numbers, so doing them once gives the same bits */
VRC_HD void vrc_composite( vrc_f2& c, const vrc_f2& e, bool frozen = false )
{
    VRC_FAST_FP
    float t = 1.0f - c.w;
    t = frozen ? 0.0f : t;
    c.x = c.x + e.x * t;
    c.w = c.w + e.w * t;
}

/* ------------------------------------------------------------------------------------------
 * March one brick segment (Renderer.cu:206-223).
 *
 * The kernel is VALU-issue bound on MI355X (a wave64 VALU instruction costs ~4 cycles of its
 * SIMD; measured: removing every memory access changes the frame time by < 15 %), so the
 * march is organised to spend as few vector instructions per sample as possible:
 *   - samples are taken in groups of VRC_GROUP: indices, then VRC_GROUP byte gathers issued
 *     back to back, then VRC_GROUP table reads, then the blends;
 *   - FAST groups: while more than VRC_GROUP steps remain on the segment every sample of the
 *     group is one the reference loop reaches, so there is no per-sample bounds test, no
 *     index select and no table-index select;
 *   - early ray termination is tested once per group (alpha never decreases); only when the
 *     threshold was crossed inside the group is the group replayed sample by sample from the
 *     saved colour, which reproduces the reference's exit after the crossing sample exactly;
 *   - the TAIL (at most VRC_GROUP + 1 remaining steps) runs the general branch-free form: a
 *     sample the reference would not reach fetches element 0 and blends table entry 256,
 *     which is all zeros -- an exact no-op.
 * The composited sample sequence is the reference's, sample for sample, in both paths.
 * lut has 257 entries; lut[256] = 0.
 * Returns true when the early-ray-termination threshold was crossed (Renderer.cu:219-226).
 * ---------------------------------------------------------------------------------------- */
#ifndef VRC_GROUP
#define VRC_GROUP 8
#endif

struct vrc_classifier;
VRC_HD vrc_f4 vrc_classify( const vrc_f4* tfp, float d, const vrc_classifier& k );
VRC_HD vrc_f2 vrc_classify( const vrc_f2* tfp, float d, const vrc_classifier& k );

/* table entry of a fetched voxel: lut[d] -- or, PERSAMPLE (volumes the 257-entry table cannot index: 16-bit
 * voxels), the transfer function and the opacity correction evaluated on the value (vrc_classify; lut is then the
 * padded transfer function and the entries are four floats) */
template < bool PERSAMPLE, typename E >
VRC_HD E vrc_entry( const E* lut, uint32_t d, const vrc_classifier* cls )
{
    if constexpr( PERSAMPLE )
        return vrc_classify( lut, (float)d, *cls );
    else
        return lut[d];
}

template < bool CLAMP, bool COUNT, bool FIXED, typename ATLAS_T, int GROUP, typename E, bool PERSAMPLE = false >
VRC_HD bool vrc_march_segment_as( const vrc_frame& f, const vrc_dev_node& n, vrc_segment s,
                                  const ATLAS_T* __restrict__ atlas, const E* lut,
                                  E& color, uint32_t& nSamples, float levelStep,
                                  const vrc_classifier* cls = nullptr )
{
    /* levelStep: step of a coarser brick under per-ray LOD (vrc_pixel_ray_lod); 0 = the frame's */
    const float stepSize = levelStep > 0.0f ? levelStep : f.stepSize;
    const vrc_sampler sm = vrc_make_sampler( n, f );
    float travel = s.dist;
    vrc_f3 pos = s.pos;
    bool done = false;
    if( !( travel > 0.0f ) )
        return false;
    vrc_fixpos fp = { 0, 0, 0, 0, 0, 0 };
    if( FIXED )
        fp = vrc_fixpos_init( sm, pos, s.step );

    /* all GROUP samples of a group are reached by the reference loop if more than
     * GROUP steps remain (the sequentially rounded travel differs from the exact one by
     * far less than one step) */
    const float guard = stepSize * (float)( GROUP + 1 );
#if defined( VRC_PIPELINE ) && defined( __HIP_DEVICE_COMPILE__ )
    /* Two groups in flight: the gathers of the NEXT group are issued before the table reads and blends of the
     * current one.  Every wave's life is a chain of groups, each as long as its slowest gather (with every voxel
     * line touched once per frame, some lane of every group goes all the way to HBM); with the next group's
     * gathers already on their way that wait overlaps the current group's work instead of following it.  Same
     * samples in the same order: only when the loads are issued changes. */
    if( travel > guard )
    {
        uint32_t idx[GROUP], d[GROUP];
        if( FIXED )
            vrc_group_indices_fixed< GROUP >( sm, fp, idx );
        else
            vrc_group_indices< CLAMP, GROUP >( sm, pos, s.step, idx );
#pragma unroll
        for( int k = 0; k < GROUP; ++k )
            d[k] = (uint32_t)atlas[idx[k]];
        for( ;; )
        {
#pragma unroll
            for( int k = 0; k < GROUP; ++k )
                travel -= stepSize; /* same sequential subtraction as the reference */
            const bool next = travel > guard; /* another whole group follows: fetch it now */
            uint32_t dn[GROUP];
#pragma unroll
            for( int k = 0; k < GROUP; ++k )
                dn[k] = 0u;
            if( next )
            {
                if( FIXED )
                    vrc_group_indices_fixed< GROUP >( sm, fp, idx );
                else
                    vrc_group_indices< CLAMP, GROUP >( sm, pos, s.step, idx );
#pragma unroll
                for( int k = 0; k < GROUP; ++k )
                    dn[k] = (uint32_t)atlas[idx[k]];
            }
            E e[GROUP];
#pragma unroll
            for( int k = 0; k < GROUP; ++k )
                e[k] = lut[d[k]];
            const E saved = color;
#pragma unroll
            for( int k = 0; k < GROUP; ++k )
                vrc_composite( color, e[k] );
            if( COUNT )
                nSamples += GROUP;
            if( color.w > VRC_EARLY_EXIT )
            {
                /* crossed inside this group: replay it with the reference's per-sample exit */
                color = saved;
                if( COUNT )
                    nSamples -= GROUP;
#pragma unroll
                for( int k = 0; k < GROUP; ++k )
                {
                    vrc_composite( color, e[k], done );
                    if( COUNT )
                        nSamples += done ? 0u : 1u;
                    done = done || ( color.w > VRC_EARLY_EXIT );
                }
                return true;
            }
            if( !next )
                break;
#pragma unroll
            for( int k = 0; k < GROUP; ++k )
                d[k] = dn[k];
        }
    }
#endif
#if defined( VRC_ZRUN )
    uint32_t zrunTag = 0xFFFFFFFFu, zrunWord = 0u;
#endif
#if defined( VRC_INT_STEPS ) /* developer measurement (round 3): the fast groups counted in integers.
                              * Only equal to the reference's float chain when the step is a power of two (every
                              * subtraction is then exact); measured on C2, where it is (1/1024): see DESIGN.md section 4 */
    uint32_t fastLeft = (uint32_t)( travel / stepSize ); /* exact for a power-of-two step */
    float travelDone = 0.0f;
    while( fastLeft > (uint32_t)( GROUP + 1 ) )
#else
    while( travel > guard )
#endif
    {
        uint32_t idx[GROUP];
        if( FIXED )
            vrc_group_indices_fixed< GROUP >( sm, fp, idx );
        else
            vrc_group_indices< CLAMP, GROUP >( sm, pos, s.step, idx );
#if defined( VRC_INT_STEPS )
        fastLeft -= (uint32_t)GROUP;
        travelDone += stepSize * (float)GROUP;
#else
#pragma unroll
        for( int k = 0; k < GROUP; ++k )
            travel -= stepSize; /* same sequential subtraction as the reference */
#endif
#if defined( VRC_SETPRIO ) && defined( __HIP_DEVICE_COMPILE__ ) /* developer measurement */
        __builtin_amdgcn_s_setprio( 3 );
#endif
        E e[GROUP];
#if defined( VRC_ZRUN ) && defined( __HIP_DEVICE_COMPILE__ ) && VRC_LAYOUT == 5
        /* developer experiment (tools/dev_layouts.sh): a lane keeps the dword of its voxel
         * column (four voxels along z) and loads a new one only when its dword address changes; the other lanes'
         * loads are out of range of the buffer descriptor and touch nothing */
        if( sizeof( ATLAS_T ) == 1 )
        {
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast< ATLAS_T* >( atlas ), (short)0, (int)0xFFFFFFFEu, 0x00020000 );
            uint32_t w[GROUP];
            uint32_t tag = zrunTag;
#pragma unroll
            for( int k = 0; k < GROUP; ++k )
            {
                const uint32_t t = idx[k] >> 2;
                const bool need = t != tag;
                tag = t;
                w[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32( rsrc, need ? (int32_t)( idx[k] & ~3u ) : -1, 0, 0 );
                idx[k] = ( idx[k] & 3u ) | ( need ? 4u : 0u );
            }
            zrunTag = tag;
#pragma unroll
            for( int k = 0; k < GROUP; ++k )
            {
                zrunWord = ( idx[k] & 4u ) ? w[k] : zrunWord;
                e[k] = lut[( zrunWord >> ( ( idx[k] & 3u ) * 8u ) ) & 0xFFu];
            }
        }
        else
#endif
#pragma unroll
        for( int k = 0; k < GROUP; ++k )
        {
#if defined( VRC_ABLATE_NO_FETCH ) /* timing experiment only */
            e[k] = lut[64u + ( idx[k] >> 31 )];
#elif defined( VRC_ABLATE_HALF_FETCH ) /* timing experiment only: every other gather dropped */
            e[k] = lut[(uint32_t)atlas[idx[k & ~1]] + ( idx[k] >> 31 )];
#elif defined( VRC_ABLATE_QUARTER_FETCH ) /* timing experiment only */
            e[k] = lut[(uint32_t)atlas[idx[k & ~3]] + ( idx[k] >> 31 )];
#else
            e[k] = vrc_entry< PERSAMPLE, E >( lut, (uint32_t)atlas[idx[k]], cls );
#endif
        }
#if defined( VRC_SETPRIO ) && defined( __HIP_DEVICE_COMPILE__ )
        __builtin_amdgcn_s_setprio( 0 );
#endif
        const E saved = color;
#pragma unroll
        for( int k = 0; k < GROUP; ++k )
            vrc_composite( color, e[k] );
        if( COUNT )
            nSamples += GROUP;
        if( color.w > VRC_EARLY_EXIT )
        {
            /* crossed inside this group: replay it with the reference's per-sample exit */
            color = saved;
            if( COUNT )
                nSamples -= GROUP;
#pragma unroll
            for( int k = 0; k < GROUP; ++k )
            {
                vrc_composite( color, e[k], done );
                if( COUNT )
                    nSamples += done ? 0u : 1u;
                done = done || ( color.w > VRC_EARLY_EXIT );
            }
            return true;
        }
    }

#if defined( VRC_INT_STEPS )
    travel -= travelDone; /* exact for a power-of-two step */
#endif
    /* tail: general form, in smaller groups (fewer slots wasted on steps no lane takes) */
#if defined( VRC_TAIL_GROUP )
    constexpr int TAILG = VRC_TAIL_GROUP;
#else
    /* measured on C2 (groups of 8): tail groups of 8 / 4 / 2 / 1: 0.500 / 0.491 / 0.484 / 0.502 ms relative:
     * fewer gather slots spent on steps no lane takes, until the dependent round trips of a short group
     * cost more than the slots did */
    /* (whole groups of GROUP / 2 between the full groups and this tail were tried for GROUP = 14: the second unrolled
     * body costs the registers of a fifth wave -- spills, 0.488 -> 0.539 ms) */
    /* groups of 14: tails of 2 / 3 / 4 / 5 / 7: 0.492 / 0.489 / 0.479 / 0.485 / 0.485 ms (C2), 0.622 / 0.615 / 0.587 / 0.580 /
     * 0.582 ms (noise, 30/20 degrees) */
    constexpr int TAILG = GROUP >= 8 ? ( GROUP + 2 ) / 4 : ( GROUP >= 2 ? GROUP / 2 : 1 );
#endif
    while( travel > 0.0f && !done )
    {
        uint32_t idx[TAILG], d[TAILG], cnt = 0;
        if( FIXED )
            vrc_group_indices_fixed< TAILG >( sm, fp, idx );
        else
            vrc_group_indices< CLAMP, TAILG >( sm, pos, s.step, idx );
#pragma unroll
        for( int k = 0; k < TAILG; ++k )
        {
            const bool v = travel > 0.0f;
            cnt += v ? 1u : 0u;
            idx[k] = v ? idx[k] : 0u;
            travel -= stepSize;
        }
#pragma unroll
        for( int k = 0; k < TAILG; ++k )
            d[k] = (uint32_t)atlas[idx[k]];
        E e[TAILG];
#pragma unroll
        for( int k = 0; k < TAILG; ++k )
        {
            if constexpr( PERSAMPLE )
            {
                /* a step the reference does not take blends nothing */
                const E z = {};
                e[k] = (uint32_t)k < cnt ? vrc_entry< true, E >( lut, d[k], cls ) : z;
            }
            else
                e[k] = lut[(uint32_t)k < cnt ? d[k] : 256u];
        }
#pragma unroll
        for( int k = 0; k < TAILG; ++k )
        {
            const bool active = ( (uint32_t)k < cnt ) && !done;
            vrc_composite( color, e[k], done );
            if( COUNT )
                nSamples += active ? 1u : 0u;
            done = done || ( color.w > VRC_EARLY_EXIT );
        }
    }
    return done;
}

template < bool CLAMP, bool COUNT, bool FIXED, typename ATLAS_T, int GROUP = VRC_GROUP, bool GREY = false >
VRC_HD bool vrc_march_segment( const vrc_frame& f, const vrc_dev_node& n, vrc_segment s,
                               const ATLAS_T* __restrict__ atlas, const vrc_f4* lut,
                               vrc_f4& color, uint32_t& nSamples, float levelStep = 0.0f )
{
    if constexpr( GREY )
    {
        /* lut holds two-float entries (the kernel fills it that way); the colour came in grey (a cleared pixel)
         * and leaves grey */
        vrc_f2 c = { color.x, color.w };
        const bool done = vrc_march_segment_as< CLAMP, COUNT, FIXED, ATLAS_T, GROUP, vrc_f2 >(
            f, n, s, atlas, reinterpret_cast< const vrc_f2* >( lut ), c, nSamples, levelStep );
        color.x = color.y = color.z = c.x;
        color.w = c.w;
        return done;
    }
    else
        return vrc_march_segment_as< CLAMP, COUNT, FIXED, ATLAS_T, GROUP, vrc_f4 >( f, n, s, atlas, lut, color,
                                                                                  nSamples, levelStep );
}

/* ------------------------------------------------------------------------------------------
 * EXTENSION: trilinear volume filter (VRC_OPT_FILTER = 1).  Not in the reference, whose 3-D
 * sampler is point-sampled (cuda/TexturePool.cu:167); named by the build's north star.  The
 * integrator is the reference's, statement for statement, with the fetch replaced by a
 * trilinear one (texel centres at i + 0.5, exact float weights, clamp addressing) -- the
 * oracle's fetch_trilinear.  The density is then continuous, so the transfer function is
 * evaluated per sample (post-classification, Renderer.cu:215-218) instead of through the
 * 257-entry classified table:
 *   tf = tex1D(tfTex, d*mult+add), alpha' = 1 - pow(1 - min(tf.a, 255/256), alphaCorrection).
 * tfp is the transfer function padded by one entry on both sides (tfp[j] = tf[clamp(j-1)],
 * 258 entries) so the two lerp operands need no index clamps.
 * ---------------------------------------------------------------------------------------- */
#define VRC_TFP_ENTRIES 258u

struct vrc_classifier
{
    float mult, add;       /* u*256 - 0.5 = d*mult + add */
    float alphaCorrection;
    float q, invq;         /* TF lerp weight quantisation (2^fracBits), q = 0: exact */
};

VRC_HD vrc_classifier vrc_make_classifier( vrc_lut_params p )
{
    vrc_classifier k;
    const float multiplyer = 1.0f / ( p.rangeMax - p.rangeMin );
    const float addedValue = -p.rangeMin / ( p.rangeMax - p.rangeMin );
    k.mult = multiplyer * 256.0f;
    k.add = addedValue * 256.0f - 0.5f;
    k.alphaCorrection = p.alphaCorrection;
    k.q = p.fracBits > 0 ? (float)( 1 << p.fracBits ) : 0.0f;
    k.invq = p.fracBits > 0 ? 1.0f / (float)( 1 << p.fracBits ) : 0.0f;
    return k;
}

#if defined( __HIP_DEVICE_COMPILE__ )
#define VRC_LERP2( B, X, A, Y ) __builtin_fmaf( ( A ), ( Y ), ( B ) * ( X ) )
#else
#define VRC_LERP2( B, X, A, Y ) ( ( B ) * ( X ) + ( A ) * ( Y ) )
#endif
VRC_HD vrc_f4 vrc_classify( const vrc_f4* tfp, float d, const vrc_classifier& k )
{
    VRC_FAST_FP
    float xB = d * k.mult + k.add;
    xB = fminf( fmaxf( xB, -1.0f ), 255.99998f );
    const float fl = floorf( xB );
    float a = xB - fl;
    if( k.q > 0.0f )
        a = floorf( a * k.q + 0.5f ) * k.invq;
    const int j = (int)fl + 1;
    const vrc_f4 t0 = tfp[j], t1 = tfp[j + 1];
    const float b = 1.0f - a;
    /* on the device the interpolation is spelled as one product and one fused multiply-add: which of the two
     * products hipcc fuses otherwise depends on the code around it, and the grey form (below) has to produce the
     * same bits as this one */
    const float cw = VRC_LERP2( b, t0.w, a, t1.w );
    const float corr = 1.0f - fminf( cw, 1.0f - 1.0f / 256.0f );
#if defined( __HIP_DEVICE_COMPILE__ )
    const float alpha = 1.0f - __builtin_amdgcn_exp2f( k.alphaCorrection * __builtin_amdgcn_logf( corr ) );
#else
    const float alpha = 1.0f - powf( corr, k.alphaCorrection );
#endif
    vrc_f4 e;
    e.x = VRC_LERP2( b, t0.x, a, t1.x ) * alpha;
    e.y = VRC_LERP2( b, t0.y, a, t1.y ) * alpha;
    e.z = VRC_LERP2( b, t0.z, a, t1.z ) * alpha;
    e.w = alpha;
    return e;
}

/* the same for a grey transfer function kept as (grey, alpha) pairs: the colour channels of vrc_classify are one
 * expression evaluated three times on equal numbers */
VRC_HD vrc_f2 vrc_classify( const vrc_f2* tfp, float d, const vrc_classifier& k )
{
    VRC_FAST_FP
    float xB = d * k.mult + k.add;
    xB = fminf( fmaxf( xB, -1.0f ), 255.99998f );
    const float fl = floorf( xB );
    float a = xB - fl;
    if( k.q > 0.0f )
        a = floorf( a * k.q + 0.5f ) * k.invq;
    const int j = (int)fl + 1;
    const vrc_f2 t0 = tfp[j], t1 = tfp[j + 1];
    const float b = 1.0f - a;
    const float cw = VRC_LERP2( b, t0.w, a, t1.w );
    const float corr = 1.0f - fminf( cw, 1.0f - 1.0f / 256.0f );
#if defined( __HIP_DEVICE_COMPILE__ )
    const float alpha = 1.0f - __builtin_amdgcn_exp2f( k.alphaCorrection * __builtin_amdgcn_logf( corr ) );
#else
    const float alpha = 1.0f - powf( corr, k.alphaCorrection );
#endif
    vrc_f2 e;
    e.x = VRC_LERP2( b, t0.x, a, t1.x ) * alpha;
    e.w = alpha;
    return e;
}

/* the eight taps of one sample: per-axis address parts (micro-blocked slot layout, see
 * vrc_voxel_address) of floor(c) and floor(c)+1, and the weights */
struct vrc_taps
{
    uint32_t ax[2], ay[2], az[2];
    float wx, wy, wz;
};

template < bool CLAMP >
VRC_HD vrc_taps vrc_trilinear_taps( const vrc_sampler& s, float lx, float ly, float lz )
{
    vrc_taps t;
    /* texel centres at i + 0.5 */
    float cx = lx - 0.5f, cy = ly - 0.5f, cz = lz - 0.5f;
    const float fx = floorf( cx ), fy = floorf( cy ), fz = floorf( cz );
    t.wx = cx - fx;
    t.wy = cy - fy;
    t.wz = cz - fz;
    int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    if( CLAMP )
    {
        const int hx = (int)s.hix, hy = (int)s.hiy, hz = (int)s.hiz;
        x0 = x0 < 0 ? 0 : ( x0 > hx ? hx : x0 );
        x1 = x1 < 0 ? 0 : ( x1 > hx ? hx : x1 );
        y0 = y0 < 0 ? 0 : ( y0 > hy ? hy : y0 );
        y1 = y1 < 0 ? 0 : ( y1 > hy ? hy : y1 );
        z0 = z0 < 0 ? 0 : ( z0 > hz ? hz : z0 );
        z1 = z1 < 0 ? 0 : ( z1 > hz ? hz : z1 );
    }
    const uint32_t ux[2] = { (uint32_t)x0, (uint32_t)x1 }, uy[2] = { (uint32_t)y0, (uint32_t)y1 },
                   uz[2] = { (uint32_t)z0, (uint32_t)z1 };
#pragma unroll
    for( int i = 0; i < 2; ++i )
    {
        t.ax[i] = vrc_mul24( ux[i] >> VRC_MB_SHIFT, 504u ) + ux[i];
        t.ay[i] = vrc_mul24( uy[i] >> VRC_MB_SHIFT, s.cyy ) + ( uy[i] << 3 ) + VRC_MB_FIX_Y( uy[i] );
        t.az[i] = vrc_mul24( uz[i] >> VRC_MB_SHIFT, s.czz ) + ( uz[i] << 6 ) - VRC_MB_FIX_Z( uz[i] ) + s.slotBase;
    }
    return t;
}

VRC_HD float vrc_trilerp( const float v[8], float wx, float wy, float wz )
{
    VRC_FAST_FP
    /* the oracle's order: x, then y, then z; a*(1-w) + b*w */
    const float ux = 1.0f - wx, uy = 1.0f - wy, uz = 1.0f - wz;
    const float c00 = v[0] * ux + v[1] * wx;
    const float c10 = v[2] * ux + v[3] * wx;
    const float c01 = v[4] * ux + v[5] * wx;
    const float c11 = v[6] * ux + v[7] * wx;
    const float c0 = c00 * uy + c10 * wy;
    const float c1 = c01 * uy + c11 * wy;
    return c0 * uz + c1 * wz;
}

/* one voxel by its element index.  On the device the index goes through an empty asm and the pointer is typed as
 * global memory: the load then takes the scalar base + 32-bit lane offset form where the base is wave-uniform and
 * the voxels are bytes; written as atlas[valid ? e : 0u] the compiler widened a select of 64-bit values and spent a
 * 64-bit vector addition per gather (eight per trilinear sample). */
template < typename ATLAS_T >
VRC_HD ATLAS_T vrc_gather( const ATLAS_T* __restrict__ atlas, uint32_t e )
{
#if defined( __HIP_DEVICE_COMPILE__ )
    typedef __attribute__( ( address_space( 1 ) ) ) const ATLAS_T g_t;
    asm( "" : "+v"( e ) );
    return ( (g_t*)atlas )[e];
#else
    return atlas[e];
#endif
}

/* gather form of the trilinear march: eight byte gathers per sample, groups of
 * VRC_LGROUP samples, general branch-free form throughout (a sample the reference loop would
 * not reach reads element 0 and is blended with weight 0).  Sample positions follow the
 * reference's float accumulation.  The LDS-staged kernel (vrc_kernels.hip) is the fast
 * path for overlap >= 1; this form also serves the clamped sampler. */
#ifndef VRC_LGROUP
#define VRC_LGROUP 4
#endif

template < bool CLAMP, bool COUNT, bool TRILINEAR, typename ATLAS_T >
VRC_HD bool vrc_march_segment_linear( const vrc_frame& f, const vrc_dev_node& n, vrc_segment s,
                                      const ATLAS_T* __restrict__ atlas, const vrc_f4* tfp,
                                      const vrc_classifier& cls, vrc_f4& color,
                                      uint32_t& nSamples, float levelStep = 0.0f )
{
    const float stepSize = levelStep > 0.0f ? levelStep : f.stepSize;
    const vrc_sampler sm = vrc_make_sampler( n, f );
    float travel = s.dist;
    vrc_f3 pos = s.pos;
    bool done = false;
    while( travel > 0.0f && !done )
    {
        bool valid[VRC_LGROUP];
        float d[VRC_LGROUP];
        if( TRILINEAR )
        {
            vrc_taps t[VRC_LGROUP];
#pragma unroll
            for( int k = 0; k < VRC_LGROUP; ++k )
            {
                VRC_FAST_FP
                valid[k] = travel > 0.0f;
                const float lx = ( pos.x - sm.minx ) * sm.kx + sm.ox;
                const float ly = ( pos.y - sm.miny ) * sm.ky + sm.oy;
                const float lz = ( pos.z - sm.minz ) * sm.kz + sm.oz;
                t[k] = vrc_trilinear_taps< CLAMP >( sm, lx, ly, lz );
                pos.x += s.step.x;
                pos.y += s.step.y;
                pos.z += s.step.z;
                travel -= stepSize;
            }
            float v[VRC_LGROUP][8];
#pragma unroll
            for( int k = 0; k < VRC_LGROUP; ++k )
#pragma unroll
                for( int c = 0; c < 8; ++c )
                {
                    const uint32_t e = t[k].ax[c & 1] + t[k].ay[( c >> 1 ) & 1] + t[k].az[c >> 2];
                    v[k][c] = (float)vrc_gather( atlas, valid[k] ? e : 0u );
                }
#pragma unroll
            for( int k = 0; k < VRC_LGROUP; ++k )
                d[k] = vrc_trilerp( v[k], t[k].wx, t[k].wy, t[k].wz );
        }
        else
        {
            /* point sampling of a volume the classified table cannot index (16-bit voxels) */
            uint32_t idx[VRC_LGROUP];
#pragma unroll
            for( int k = 0; k < VRC_LGROUP; ++k )
            {
                valid[k] = travel > 0.0f;
                travel -= stepSize;
            }
            vrc_group_indices< CLAMP, VRC_LGROUP >( sm, pos, s.step, idx );
#pragma unroll
            for( int k = 0; k < VRC_LGROUP; ++k )
                d[k] = (float)vrc_gather( atlas, valid[k] ? idx[k] : 0u );
        }
#pragma unroll
        for( int k = 0; k < VRC_LGROUP; ++k )
        {
            const vrc_f4 e = vrc_classify( tfp, d[k], cls );
            const bool active = valid[k] && !done;
            vrc_composite( color, e, !active );
            if( COUNT )
                nSamples += active ? 1u : 0u;
            done = done || ( active && color.w > VRC_EARLY_EXIT );
        }
    }
    return done;
}

/* ------------------------------------------------------------------------------------------
 * EXTENSION, tap-packed form of the trilinear filter (round 4).  The filter the north star names is one sampler
 * enum in the reference (cuda/TexturePool.cu:163-170, cudaFilterModePoint -> Linear) and free in its hardware; here
 * the eight taps were eight byte gathers (above) or a staged box in LDS (vrc_kernels_lds.hip), both bound by the
 * instructions and cache look-ups around the sample while HBM idles.  This form spends memory instead: next to the
 * byte atlas the pool keeps a second atlas of 16-bit texels, each a voxel and its neighbour along z,
 *     t(x,y,z) = v[x,y,z] | v[x,y,z+1] << 8      (16-bit voxels: 32-bit texels, << 16, the gathers 8 bytes each),
 * in which t(x,y,z) and t(x+1,y,z) are ALWAYS neighbours in memory: the texels lie in blocks of 8x8x8 whose x-rows
 * carry a ninth texel, a copy of the next block's first (9/8 x 2 = 2.25 times the bytes of the byte atlas; written by
 * vrc_k_pack_slots when a brick is uploaded; the slot's overlap >= 1 supplies the +1 neighbours).  The eight taps of
 * a sample are then TWO 4-byte gathers -- global_load_dword at 2-byte-aligned addresses that never leave an 18-byte
 * row: t(x0, y0, z0) | t(x0+1, y0, z0) << 16 and the same at y0 + 1 -- eight byte -> float conversions straight out
 * of the two registers and the seven interpolations: no box, no staging, no walk batching, the same work for every
 * view direction.  The 64 rows of a block lie in Morton order of (y, z), so a 128-byte line holds 9 x 2 x 2 ... 9 x 4
 * x 2 texels and what a tile touches per step is about the same along every axis.
 * (Measured and dropped, profiles/r4_packed_kernel_experiments.txt: 32-bit texels of the 2x2 neighbourhood in (x, y)
 * or (y, z) -- one 8-byte gather, twice the line fills --, rows of 8 with the pair across a block fetched by masked
 * second gathers, and four other row orders.)
 *
 * Sample positions, weights, interpolation and classification are the staged kernel's, operation for operation
 * (8.24 fixed-point positions from vrc_fixpos_init minus half a voxel, 24-bit weights used unscaled, x then y then
 * z, the transfer function through CUDA's 1.8 fixed-point weight out of one float -> integer conversion), so the
 * two forms composite the same numbers.
 * ---------------------------------------------------------------------------------------- */
/* slot-local element order of the packed atlas: offset = PX(x) + PY(y) + PZ(z), blocks x-fastest inside the slot */
/* bytes per texel: twice the voxel's -- 2 (8-bit voxels: TB = 2 below) or 4 (16-bit voxels: t = v[z] | v[z+1] << 16,
 * 4.5 bytes per voxel next to the atlas's 2; the pairs of rows y and y + 1 are then two 8-byte gathers) */
#define VRC_PK_TEXEL( voxelBytes ) ( 2u * ( voxelBytes ) )
#define VRC_PK_ROW 9u
#define VRC_PK_BLOCK ( VRC_PK_ROW * 64u ) /* 576 texels */
/* row number of (y & 7, z & 7) inside a block: the bits of y and z interleaved */
VRC_HD uint32_t vrc_pk_ry( uint32_t u ) { return ( u & 1u ) | ( ( ( u >> 1 ) & 1u ) << 2 ) | ( ( ( u >> 2 ) & 1u ) << 4 ); }
VRC_HD uint32_t vrc_pk_rz( uint32_t u ) { return vrc_pk_ry( u ) << 1; }
VRC_HD uint32_t vrc_pk_x( uint32_t u ) { return ( u & 7u ) + VRC_PK_BLOCK * ( u >> 3 ); }
VRC_HD uint32_t vrc_pk_y( uint32_t u, uint32_t sbx ) { return VRC_PK_ROW * vrc_pk_ry( u ) + VRC_PK_BLOCK * sbx * ( u >> 3 ); }
VRC_HD uint32_t vrc_pk_z( uint32_t u, uint32_t sbx, uint32_t sby ) { return VRC_PK_ROW * vrc_pk_rz( u ) + VRC_PK_BLOCK * sbx * sby * ( u >> 3 ); }
VRC_HD uint32_t vrc_packed_local_index( uint32_t x, uint32_t y, uint32_t z, uint32_t sbx, uint32_t sby )
{
    return vrc_pk_x( x ) + vrc_pk_y( y, sbx ) + vrc_pk_z( z, sbx, sby );
}
/* slot-local voxel of packed element `in` of a block (the pack kernel's decode); ix == 8: the copy's column */
VRC_HD void vrc_packed_decode( uint32_t in, uint32_t& ix, uint32_t& iy, uint32_t& iz )
{
    const uint32_t r = in / VRC_PK_ROW;
    ix = in % VRC_PK_ROW;
    iy = ( r & 1u ) | ( ( ( r >> 2 ) & 1u ) << 1 ) | ( ( ( r >> 4 ) & 1u ) << 2 );
    iz = ( ( r >> 1 ) & 1u ) | ( ( ( r >> 3 ) & 1u ) << 1 ) | ( ( ( r >> 5 ) & 1u ) << 2 );
}
/* texels of a packed slot / element offset of the packed slot that belongs to the byte slot at element slotBase
 * (byte slots are whole blocks of 512) */
VRC_HD uint64_t vrc_packed_elems( uint64_t byteElems ) { return byteElems / VRC_MB_VOXELS * VRC_PK_BLOCK; }

/* texel of a voxel and its neighbour along z; T = uint16_t (8-bit voxels) or uint32_t (16-bit voxels) */
template < typename T >
VRC_HD T vrc_pack_taps( uint32_t vz0, uint32_t vz1 ) { return (T)( vz0 | ( vz1 << ( 4u * sizeof( T ) ) ) ); }

/* classifier of a trilinear sample that arrives scaled by 2^72 (three unscaled 24-bit weights): the oracle's
 * orc_tf_fetch + composite (cuda/ColorMap.cu:40-45, cuda/Renderer.cu:83-93) with the transfer-function texel pair
 * and CUDA's 1.8 fixed-point lerp weight taken out of ONE float -> integer conversion: tq = xB * 256 + 256.5 with
 * xB = u * 256 - 0.5 the texel coordinate; texel pair (tq >> 8, + 1) of the padded table, weight (tq & 255) / 256
 * rounded to nearest as the hardware does.  VRC_OPT_TF_FRAC_BITS = 8 only. */
struct vrc_cls8
{
    float mult, add, kexp;
};
VRC_HD vrc_cls8 vrc_make_cls8( const vrc_classifier& c )
{
    vrc_cls8 k;
    k.mult = c.mult * 256.0f * 0x1p-72f;
    k.add = c.add * 256.0f + 256.5f;
    k.kexp = c.alphaCorrection;
    return k;
}
/* entries of the classifier's table: per texel j of the padded transfer function tfp (VRC_TFP_ENTRIES entries)
 * (colour_j, 1 - alpha_j) / 256 -- the weights are used as integers 0..256 -- one texel more than tfp has: a
 * sample at the upper clamp reads the pair (257, 258) with weights (256, 0).  Grey transfer function: texels j and
 * j + 1 side by side, (g_j, 1 - a_j, g_j+1, 1 - a_j+1) / 256, one 16-byte read per sample. */
#define VRC_CLS8_ENTRIES ( VRC_TFP_ENTRIES + 1u )
VRC_HD vrc_f4 vrc_cls8_entry( const vrc_f4* tfp, uint32_t i, bool grey )
{
    const float sc = 1.0f / 256.0f;
    const vrc_f4 e0 = tfp[i < VRC_TFP_ENTRIES - 1u ? i : VRC_TFP_ENTRIES - 1u];
    const vrc_f4 e1 = tfp[i + 1u < VRC_TFP_ENTRIES - 1u ? i + 1u : VRC_TFP_ENTRIES - 1u];
    vrc_f4 t;
    if( grey )
    {
        t.x = e0.x * sc; t.y = ( 1.0f - e0.w ) * sc; t.z = e1.x * sc; t.w = ( 1.0f - e1.w ) * sc;
    }
    else
    {
        t.x = e0.x * sc; t.y = e0.y * sc; t.z = e0.z * sc; t.w = ( 1.0f - e0.w ) * sc;
    }
    return t;
}
VRC_HD float vrc_alpha8( float corr, float kexp )
{
    /* 1 - min(a, 255/256) = max(1 - a, 1/256) (Renderer.cu:88); pow as exp2(k log2 x) on the device */
    corr = fmaxf( corr, 1.0f / 256.0f );
#if defined( __HIP_DEVICE_COMPILE__ )
    return 1.0f - __builtin_amdgcn_exp2f( kexp * __builtin_amdgcn_logf( corr ) );
#else
    return 1.0f - powf( corr, kexp );
#endif
}
VRC_HD uint32_t vrc_cls8_texel( float d, const vrc_cls8& k )
{
    float tq = __builtin_fmaf( d, k.mult, k.add );
#if defined( __HIP_DEVICE_COMPILE__ )
    tq = __builtin_amdgcn_fmed3f( tq, 0.5f, 65792.25f );
#else
    tq = fminf( fmaxf( tq, 0.5f ), 65792.25f );
#endif
    return (uint32_t)tq;
}
VRC_HD vrc_f2 vrc_classify8( const vrc_f2*, const vrc_f4* tab, float d, const vrc_cls8& k )
{
    const uint32_t u = vrc_cls8_texel( d, k );
    const float a = (float)( u & 255u ), b = 256.0f - a;
    const vrc_f4 t = *reinterpret_cast< const vrc_f4* >( reinterpret_cast< const char* >( tab ) + ( ( u >> 4 ) & 0xFFFF0u ) );
    const float alpha = vrc_alpha8( __builtin_fmaf( a, t.w, b * t.y ), k.kexp );
    vrc_f2 e;
    e.x = __builtin_fmaf( a, t.z, b * t.x ) * alpha;
    e.w = alpha;
    return e;
}
VRC_HD vrc_f4 vrc_classify8( const vrc_f4*, const vrc_f4* tab, float d, const vrc_cls8& k )
{
    const uint32_t u = vrc_cls8_texel( d, k );
    const float a = (float)( u & 255u ), b = 256.0f - a;
    const vrc_f4* const p = reinterpret_cast< const vrc_f4* >( reinterpret_cast< const char* >( tab ) + ( ( u >> 4 ) & 0xFFFF0u ) );
    const vrc_f4 t0 = p[0], t1 = p[1];
    const float alpha = vrc_alpha8( __builtin_fmaf( a, t1.w, b * t0.w ), k.kexp );
    vrc_f4 e;
    e.x = __builtin_fmaf( a, t1.x, b * t0.x ) * alpha;
    e.y = __builtin_fmaf( a, t1.y, b * t0.y ) * alpha;
    e.z = __builtin_fmaf( a, t1.z, b * t0.z ) * alpha;
    e.w = alpha;
    return e;
}

/* the taps of a sample as they come out of the two gathers (TB = bytes per texel).  TB = 2: t0 = t(x0,y0,z0) |
 * t(x0+1,y0,z0) << 16, t1 the same at y0 + 1 (a texel's low byte is z0, its high byte z0 + 1).  TB = 4: one texel per
 * word, a = row y0, b = row y0 + 1, 0 / 1 = x0 / x0 + 1 (low half z0, high half z0 + 1). */
template < int TB >
struct vrc_pk_taps;
template <>
struct vrc_pk_taps< 2 >
{
    uint32_t t0, t1;
};
template <>
struct vrc_pk_taps< 4 >
{
    uint32_t a0, a1, b0, b1;
};

/* the interpolated density of a sample, times 2^72; weights = the 24 fraction bits of the sample's 8.24 coordinates,
 * NOT scaled by 2^-24 -- W and 2^24 - W are exact, so every product and sum is 2^24 (2^48, 2^72) times the one with
 * scaled weights, bit for bit; vrc_cls8.mult carries the 2^-72.  The oracle's order: x, then y, then z;
 * a * (1 - w) + b * w. */
VRC_HD float vrc_trilerp_packed( const vrc_pk_taps< 2 >& t, uint32_t fx, uint32_t fy, uint32_t fz )
{
    const uint32_t T0 = t.t0, T1 = t.t1;
    const float wx = (float)( fx & 0xFFFFFFu ), wy = (float)( fy & 0xFFFFFFu ), wz = (float)( fz & 0xFFFFFFu );
    const float ux = 16777216.0f - wx, uy = 16777216.0f - wy, uz = 16777216.0f - wz;
    const float c00 = __builtin_fmaf( (float)( ( T0 >> 16 ) & 255u ), wx, (float)( T0 & 255u ) * ux );  /* y0 z0 */
    const float c01 = __builtin_fmaf( (float)( T0 >> 24 ), wx, (float)( ( T0 >> 8 ) & 255u ) * ux );    /* y0 z1 */
    const float c10 = __builtin_fmaf( (float)( ( T1 >> 16 ) & 255u ), wx, (float)( T1 & 255u ) * ux );  /* y1 z0 */
    const float c11 = __builtin_fmaf( (float)( T1 >> 24 ), wx, (float)( ( T1 >> 8 ) & 255u ) * ux );    /* y1 z1 */
    const float c0 = __builtin_fmaf( c10, wy, c00 * uy );
    const float c1 = __builtin_fmaf( c11, wy, c01 * uy );
    return __builtin_fmaf( c1, wz, c0 * uz );
}
VRC_HD float vrc_trilerp_packed( const vrc_pk_taps< 4 >& t, uint32_t fx, uint32_t fy, uint32_t fz )
{
    const float wx = (float)( fx & 0xFFFFFFu ), wy = (float)( fy & 0xFFFFFFu ), wz = (float)( fz & 0xFFFFFFu );
    const float ux = 16777216.0f - wx, uy = 16777216.0f - wy, uz = 16777216.0f - wz;
    const float c00 = __builtin_fmaf( (float)( t.a1 & 0xFFFFu ), wx, (float)( t.a0 & 0xFFFFu ) * ux ); /* y0 z0 */
    const float c01 = __builtin_fmaf( (float)( t.a1 >> 16 ), wx, (float)( t.a0 >> 16 ) * ux );         /* y0 z1 */
    const float c10 = __builtin_fmaf( (float)( t.b1 & 0xFFFFu ), wx, (float)( t.b0 & 0xFFFFu ) * ux ); /* y1 z0 */
    const float c11 = __builtin_fmaf( (float)( t.b1 >> 16 ), wx, (float)( t.b0 >> 16 ) * ux );         /* y1 z1 */
    const float c0 = __builtin_fmaf( c10, wy, c00 * uy );
    const float c1 = __builtin_fmaf( c11, wy, c01 * uy );
    return __builtin_fmaf( c1, wz, c0 * uz );
}

/* the two gathers of a sample at byte offsets of the lane's packed slot.  TB = 2: 4-byte loads at 2-byte-aligned
 * addresses; TB = 4: 8-byte loads at 4-byte-aligned addresses (multi-dword loads need dword alignment only).  On the
 * device the pointer is typed as global memory, so the compiler emits global_load_dword / _dwordx2. */
VRC_HD vrc_pk_taps< 2 > vrc_packed_load( const vrc_pk_taps< 2 >*, const uint8_t* slot, uint32_t byteOffset, uint32_t byteOffsetY1 )
{
#if defined( VRC_PACKED_ABLATE ) && VRC_PACKED_ABLATE == 3 /* timing experiment only: 4-byte-aligned (wrong) addresses */
    byteOffset &= ~2u;
    byteOffsetY1 &= ~2u;
#endif
#if defined( __HIP_DEVICE_COMPILE__ )
    typedef uint32_t u32_a2 __attribute__( ( aligned( 2 ) ) );
    typedef __attribute__( ( address_space( 1 ) ) ) const u32_a2 g_t;
    return vrc_pk_taps< 2 >{ *reinterpret_cast< g_t* >( (uintptr_t)( slot + byteOffset ) ),
                          *reinterpret_cast< g_t* >( (uintptr_t)( slot + byteOffsetY1 ) ) };
#else
    vrc_pk_taps< 2 > t;
    memcpy( &t.t0, slot + byteOffset, 4 );
    memcpy( &t.t1, slot + byteOffsetY1, 4 );
    return t;
#endif
}
VRC_HD vrc_pk_taps< 4 > vrc_packed_load( const vrc_pk_taps< 4 >*, const uint8_t* slot, uint32_t byteOffset, uint32_t byteOffsetY1 )
{
#if defined( __HIP_DEVICE_COMPILE__ )
    typedef uint32_t u32x2_a4 __attribute__( ( ext_vector_type( 2 ), aligned( 4 ) ) );
    typedef __attribute__( ( address_space( 1 ) ) ) const u32x2_a4 g_t;
    const u32x2_a4 a = *reinterpret_cast< g_t* >( (uintptr_t)( slot + byteOffset ) );
    const u32x2_a4 b = *reinterpret_cast< g_t* >( (uintptr_t)( slot + byteOffsetY1 ) );
    return vrc_pk_taps< 4 >{ a.x, a.y, b.x, b.y };
#else
    vrc_pk_taps< 4 > t;
    memcpy( &t.a0, slot + byteOffset, 8 );
    memcpy( &t.b0, slot + byteOffsetY1, 8 );
    return t;
#endif
}
/* (timing experiments only) */
VRC_HD vrc_pk_taps< 2 > vrc_pk_taps_fake( const vrc_pk_taps< 2 >*, uint32_t off ) { return vrc_pk_taps< 2 >{ off * 0x01010101u, off * 0x00010101u }; }
VRC_HD vrc_pk_taps< 4 > vrc_pk_taps_fake( const vrc_pk_taps< 4 >*, uint32_t off ) { return vrc_pk_taps< 4 >{ off * 0x01010101u, off * 0x00010101u, off, ~off }; }
VRC_HD uint32_t vrc_pk_taps_mix( const vrc_pk_taps< 2 >& t ) { return t.t0 ^ t.t1; }
VRC_HD uint32_t vrc_pk_taps_mix( const vrc_pk_taps< 4 >& t ) { return t.a0 ^ t.a1 ^ t.b0 ^ t.b1; }

/* byte offsets (slot-local + bias) of the texel pairs of the next N samples; p advances by N steps.  Device: the per-axis
 * parts from the tables in LDS (filled by the kernel with TB * vrc_pk_x / y / z). */
template < int N, int TB >
VRC_HD void vrc_packed_offsets( const vrc_sampler& s, vrc_fixpos& p, uint32_t bias, uint32_t* off, uint32_t* offY1 )
{
#if defined( __HIP_DEVICE_COMPILE__ ) && defined( VRC_ADDR_TABLES )
    (void)s;
#pragma unroll
    for( int k = 0; k < N; ++k )
    {
        const char* const t = reinterpret_cast< const char* >( vrc_addr_tab );
        const uint32_t tx = *reinterpret_cast< const uint32_t* >( t + ( ( p.x >> 22 ) & 0x3FCu ) );
        const uint32_t* const py = reinterpret_cast< const uint32_t* >( t + 1024 + ( ( p.y >> 22 ) & 0x3FCu ) );
        const uint32_t tz = *reinterpret_cast< const uint32_t* >( t + 2048 + ( ( p.z >> 22 ) & 0x3FCu ) );
        const uint32_t xz = tx + tz + bias; /* (one v_add3_u32) */
        off[k] = xz + py[0];
        offY1[k] = xz + py[1]; /* (y = 255 reads the z table's first entry: no sample with a weight has it) */
        p.x += p.dx;
        p.y += p.dy;
        p.z += p.dz;
        asm( "" : "+v"( p.x ), "+v"( p.y ), "+v"( p.z ) ); /* keep the additions (vrc_group_indices_fixed) */
    }
#else
    const uint32_t sbx = ( s.cyy + 64u ) / VRC_MB_VOXELS, sby = ( s.czz + 512u ) / ( sbx * VRC_MB_VOXELS );
#pragma unroll
    for( int k = 0; k < N; ++k )
    {
        off[k] = bias + (uint32_t)TB * vrc_packed_local_index( p.x >> 24, p.y >> 24, p.z >> 24, sbx, sby );
        offY1[k] = bias + (uint32_t)TB * vrc_packed_local_index( p.x >> 24, ( ( p.y >> 24 ) + 1u ) & 255u, p.z >> 24, sbx, sby );
        p.x += p.dx;
        p.y += p.dy;
        p.z += p.dz;
    }
#endif
}

#ifndef VRC_PGROUP
#define VRC_PGROUP 12 /* (4 ... 24 measured: the fetches in flight per wave decide; 12 at three waves per SIMD) */
#endif
#ifndef VRC_PGROUP16
#define VRC_PGROUP16 12 /* 16-bit voxels: four registers of taps per sample in flight (6 / 8 / 12 measured: 2.03 / 1.92 / 1.80 ms on C2) */
#endif


/* March one brick segment through the packed atlas (Renderer.cu:206-223 with the trilinear fetch).  Organised as
 * vrc_march_segment_as: whole groups without per-sample tests while more than GROUP steps remain, the early-exit
 * test once per group with an exact replay, a general tail.  tab: vrc_cls8_entry table.  E: vrc_f4, or vrc_f2
 * for a grey transfer function (vrc_raycast_args.greyTable). */
template < bool COUNT, int GROUP, typename E, bool WIDE, int TB >
VRC_HD bool vrc_march_segment_packed( const vrc_frame& f, const vrc_dev_node& n, vrc_segment s,
                                      const void* __restrict__ packed, const vrc_f4* tab, const vrc_cls8& kc,
                                      E& color, uint32_t& nSamples, float levelStep )
{
    const float stepSize = levelStep > 0.0f ? levelStep : f.stepSize;
    float travel = s.dist;
    if( !( travel > 0.0f ) )
        return false;
    const vrc_sampler sm = vrc_make_sampler( n, f );
    /* the lane's packed slot.  A packed atlas of at most 4 GiB: the atlas pointer (uniform: a scalar base) + a 32-bit
     * byte offset per lane, the slot's offset folded into it.  WIDE (a larger one): a 64-bit pointer per lane. */
    const uint64_t slotBytes = vrc_packed_elems( WIDE ? ( (uint64_t)n.slotBaseHi << 32 ) | n.slotBase : (uint64_t)n.slotBase ) * (uint32_t)TB;
    const uint8_t* const slot = reinterpret_cast< const uint8_t* >( packed ) + ( WIDE ? slotBytes : 0u );
    const uint32_t bias = WIDE ? 0u : (uint32_t)slotBytes;
    vrc_fixpos fp = vrc_fixpos_init( sm, s.pos, s.step );
    /* texel centres at i + 0.5: the integer part of (coordinate - 0.5) is the lower tap, its fraction the weight */
    fp.x -= 1u << 23;
    fp.y -= 1u << 23;
    fp.z -= 1u << 23;
    bool done = false;
    const float guard = stepSize * (float)( GROUP + 1 );
    while( travel > guard )
    {
        uint32_t off[GROUP], offY1[GROUP];
        vrc_pk_taps< TB > t[GROUP];
        vrc_fixpos q = fp; /* the group's first sample: the weights are taken again from here after the loads */
        vrc_packed_offsets< GROUP, TB >( sm, fp, bias, off, offY1 );
#pragma unroll
        for( int k = 0; k < GROUP; ++k )
        {
#if defined( VRC_PACKED_ABLATE ) && VRC_PACKED_ABLATE == 1 /* timing experiment only: no fetch */
            t[k] = vrc_pk_taps_fake( (const vrc_pk_taps< TB >*)nullptr, off[k] );
#else
            t[k] = vrc_packed_load( (const vrc_pk_taps< TB >*)nullptr, slot, off[k], offY1[k] );
#endif
        }
#pragma unroll
        for( int k = 0; k < GROUP; ++k )
            travel -= stepSize; /* same sequential subtraction as the reference */
        E e[GROUP];
#if defined( VRC_PACKED_ABLATE ) && VRC_PACKED_ABLATE == 2 /* timing experiment only: fetches, next to no arithmetic */
#pragma unroll
        for( int k = 0; k < GROUP; ++k )
        {
            e[k] = E{};
            e[k].w = (float)( vrc_pk_taps_mix( t[k] ) >> 31 ) * 1e-9f;
        }
#else
#pragma unroll
        for( int k = 0; k < GROUP; ++k )
        {
            e[k] = vrc_classify8( (const E*)nullptr, tab, vrc_trilerp_packed( t[k], q.x, q.y, q.z ), kc );
            q.x += q.dx;
            q.y += q.dy;
            q.z += q.dz;
        }
#endif
        const E saved = color;
#pragma unroll
        for( int k = 0; k < GROUP; ++k )
            vrc_composite( color, e[k] );
        if( COUNT )
            nSamples += GROUP;
        if( color.w > VRC_EARLY_EXIT )
        {
            /* crossed inside this group: replay it with the reference's per-sample exit */
            color = saved;
            if( COUNT )
                nSamples -= GROUP;
#pragma unroll
            for( int k = 0; k < GROUP; ++k )
            {
                vrc_composite( color, e[k], done );
                if( COUNT )
                    nSamples += done ? 0u : 1u;
                done = done || ( color.w > VRC_EARLY_EXIT );
            }
            return true;
        }
    }
    constexpr int TAILG = GROUP >= 4 ? GROUP / 2 : 1;
    while( travel > 0.0f && !done )
    {
        uint32_t off[TAILG], offY1[TAILG], cnt = 0;
        vrc_pk_taps< TB > t[TAILG];
        vrc_fixpos q = fp;
        vrc_packed_offsets< TAILG, TB >( sm, fp, bias, off, offY1 );
#pragma unroll
        for( int k = 0; k < TAILG; ++k )
        {
            const bool v = travel > 0.0f;
            cnt += v ? 1u : 0u;
            /* a step the reference does not take reads the slot's first texels and blends nothing */
            off[k] = v ? off[k] : bias;
            offY1[k] = v ? offY1[k] : bias;
            t[k] = vrc_packed_load( (const vrc_pk_taps< TB >*)nullptr, slot, off[k], offY1[k] );
            travel -= stepSize;
        }
#pragma unroll
        for( int k = 0; k < TAILG; ++k )
        {
            const E z = {};
            const E c = vrc_classify8( (const E*)nullptr, tab, vrc_trilerp_packed( t[k], q.x, q.y, q.z ), kc );
            const bool active = ( (uint32_t)k < cnt ) && !done;
            vrc_composite( color, (uint32_t)k < cnt ? c : z, done );
            if( COUNT )
                nSamples += active ? 1u : 0u;
            done = done || ( color.w > VRC_EARLY_EXIT );
            q.x += q.dx;
            q.y += q.dy;
            q.z += q.dz;
        }
    }
    return done;
}

/* ------------------------------------------------------------------------------------------
 * Reference-order pixel: the O(nodeCount) loop of Renderer.cu:172-227, nodes in host order.
 * ---------------------------------------------------------------------------------------- */
/* MODE: how a sample is fetched and classified.
 *   VRC_MODE_TABLE      point sample of a u8 volume through the classified table (257 entries):
 *                       the reference path;
 *   VRC_MODE_TRILINEAR  trilinear fetch, per-sample classification (lut = padded transfer function);
 *   VRC_MODE_POINT      point sample, per-sample classification (16-bit voxels). */
#define VRC_MODE_GREY 3 /* VRC_MODE_TABLE with a grey transfer function (r == g == b in every entry): two-float table
                         * entries and colours, bit-identical frames, half the table bytes and blend work */
#define VRC_MODE_POINT_GREY 4 /* VRC_MODE_POINT with a grey transfer function, as VRC_MODE_GREY is to VRC_MODE_TABLE */
#define VRC_MODE_TABLE 0
#define VRC_MODE_TRILINEAR 1
#define VRC_MODE_POINT 2
#define VRC_MODE_PACKED 5      /* trilinear through the tap-packed atlas (vrc_march_segment_packed): ATLAS_T = uint32_t,
                                * lut = the vrc_cls8_entry table */
#define VRC_MODE_PACKED_GREY 6 /* the same for a grey transfer function: (grey, alpha) colours, paired table entries */

/* BIG: the atlas holds more than 2^32 voxels.  Offsets inside a slot stay 32-bit; the slot's
 * 64-bit base moves into the lane's atlas pointer (one 64-bit add per gather instead of a 32-bit
 * one), so the default kernels keep their scalar base + 32-bit offset addressing. */
template < bool CLAMP, bool COUNT, bool FIXED, int MODE, typename ATLAS_T, int GROUP = VRC_GROUP, bool BIG = false >
VRC_HD bool vrc_march_brick( const vrc_frame& f, const vrc_dev_node& n, const vrc_segment& s,
                             const ATLAS_T* __restrict__ atlas, const vrc_f4* lut,
                             const vrc_classifier& cls, vrc_f4& color, uint32_t& nSamples,
                             float levelStep = 0.0f )
{
    if constexpr( MODE == VRC_MODE_PACKED || MODE == VRC_MODE_PACKED_GREY )
    {
        /* (BIG here: a packed atlas of more than 4 GiB, or of an atlas of more than 2^32 voxels -- 64-bit lane pointers.
         * ATLAS_T is a tag: uint32_t = the packed form of 8-bit voxels, uint64_t = of 16-bit voxels) */
        static_assert( ( sizeof( ATLAS_T ) == 4 || sizeof( ATLAS_T ) == 8 ) && !CLAMP, "the packed atlas: overlap >= 1" );
        constexpr int TB = sizeof( ATLAS_T ) == 8 ? 4 : 2;
        const vrc_cls8 kc = vrc_make_cls8( cls );
        if constexpr( MODE == VRC_MODE_PACKED_GREY )
        {
            vrc_f2 c = { color.x, color.w };
            const bool done = vrc_march_segment_packed< COUNT, GROUP, vrc_f2, BIG, TB >( f, n, s, atlas, lut, kc, c, nSamples, levelStep );
            color.x = color.y = color.z = c.x;
            color.w = c.w;
            return done;
        }
        else
            return vrc_march_segment_packed< COUNT, GROUP, vrc_f4, BIG, TB >( f, n, s, atlas, lut, kc, color, nSamples, levelStep );
    }
    else if( BIG )
    {
        vrc_dev_node local = n;
        local.slotBase = 0u;
        const ATLAS_T* slot = atlas + ( ( (uint64_t)n.slotBaseHi << 32 ) | n.slotBase );
        return vrc_march_brick< CLAMP, COUNT, FIXED, MODE, ATLAS_T, GROUP, false >( f, local, s, slot, lut, cls,
                                                                                color, nSamples, levelStep );
    }
    if constexpr( MODE == VRC_MODE_GREY )
        return vrc_march_segment< CLAMP, COUNT, FIXED, ATLAS_T, GROUP, true >( f, n, s, atlas, lut, color, nSamples,
                                                                             levelStep );
    else if constexpr( MODE == VRC_MODE_POINT_GREY )
    {
        /* lut holds the padded transfer function as (grey, alpha) pairs; the pixel came in grey and leaves grey */
        vrc_f2 c = { color.x, color.w };
        const bool done = vrc_march_segment_as< CLAMP, COUNT, true, ATLAS_T, GROUP, vrc_f2, true >(
            f, n, s, atlas, reinterpret_cast< const vrc_f2* >( lut ), c, nSamples, levelStep, &cls );
        color.x = color.y = color.z = c.x;
        color.w = c.w;
        return done;
    }
    else if constexpr( MODE == VRC_MODE_POINT && FIXED )
        /* point sampling with per-sample classification (16-bit voxels) through the grouped march of the table
         * form: whole groups without per-sample masks, fixed-point stepping, address tables; only the table read
         * is replaced by vrc_classify */
        return vrc_march_segment_as< CLAMP, COUNT, true, ATLAS_T, GROUP, vrc_f4, true >( f, n, s, atlas, lut, color,
                                                                                       nSamples, levelStep, &cls );
    else if( MODE != VRC_MODE_TABLE )
        return vrc_march_segment_linear< CLAMP, COUNT, MODE == VRC_MODE_TRILINEAR, ATLAS_T >(
            f, n, s, atlas, lut, cls, color, nSamples, levelStep );
    return vrc_march_segment< CLAMP, COUNT, FIXED, ATLAS_T, GROUP >( f, n, s, atlas, lut, color, nSamples,
                                                                   levelStep );
}

/* ---- tile culling for the reference-order loop -------------------------------------------------------------
 * The reference tests every brick of the list against every ray (Renderer.cu:172-181).  A wave is an 8x8 pixel
 * tile: the bricks none of its rays can hit are found once per wave -- 64 bricks per step against the pyramid
 * through the tile's corner pixels (one pixel of margin) -- and the lanes then run the reference's loop over the
 * remaining bricks, in the list's order, with the reference's own slab test per ray.  A brick that is not hit is a
 * `continue` in the reference (the hit test comes before the `break` test, :181-186), so leaving it out changes
 * nothing: same bricks marched, same order, same samples. */
struct vrc_tile_pyramid
{
    vrc_f3 eye;
    vrc_f3 n[4]; /* inward normals of the four side planes through the eye */
};
VRC_HD vrc_f3 vrc_pixel_direction( const vrc_frame& f, float wx, float wy )
{
    const float nx = 2.0f * ( wx + f.pixelOffX - f.vpX - ( f.vpW / 2.0f ) ) / f.vpW;
    const float ny = 2.0f * ( wy + f.pixelOffY - f.vpY - ( f.vpH / 2.0f ) ) / f.vpH;
    const vrc_f4 ndc = { nx, ny, 1.0f, 1.0f };
    const vrc_f4 e = vrc_mul44( f.invProj, ndc );
    const vrc_f4 eyeSpace = { e.x / e.w, e.y / e.w, e.z / e.w, 1.0f };
    const vrc_f4 world = vrc_mul44( f.invView, eyeSpace );
    const vrc_f3 d = { world.x - f.eye[0], world.y - f.eye[1], world.z - f.eye[2] };
    return d;
}
VRC_HD vrc_f3 vrc_cross( vrc_f3 a, vrc_f3 b )
{
    const vrc_f3 c = { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
    return c;
}
/* pixels [x0, x1] x [y0, y1] (window coordinates of pixel corners, frame rows) */
VRC_HD vrc_tile_pyramid vrc_make_tile_pyramid( const vrc_frame& f, float x0, float y0, float x1, float y1 )
{
    vrc_tile_pyramid p;
    p.eye.x = f.eye[0];
    p.eye.y = f.eye[1];
    p.eye.z = f.eye[2];
    const vrc_f3 d[4] = { vrc_pixel_direction( f, x0, y0 ), vrc_pixel_direction( f, x1, y0 ),
                          vrc_pixel_direction( f, x1, y1 ), vrc_pixel_direction( f, x0, y1 ) };
    for( int k = 0; k < 4; ++k )
    {
        vrc_f3 n = vrc_cross( d[k], d[( k + 1 ) & 3] );
        if( vrc_dot( n, d[( k + 2 ) & 3] ) < 0.0f ) /* whichever way the projection winds the corners */
        {
            n.x = -n.x;
            n.y = -n.y;
            n.z = -n.z;
        }
        p.n[k] = n;
    }
    return p;
}
/* false: the box lies entirely outside one side plane, no ray of the tile can hit it */
VRC_HD bool vrc_pyramid_may_hit( const vrc_tile_pyramid& p, const float bmin[3], const float bsize[3] )
{
    const float lo[3] = { bmin[0] - p.eye.x, bmin[1] - p.eye.y, bmin[2] - p.eye.z };
    const float hi[3] = { lo[0] + bsize[0], lo[1] + bsize[1], lo[2] + bsize[2] };
    for( int k = 0; k < 4; ++k )
    {
        const float nn[3] = { p.n[k].x, p.n[k].y, p.n[k].z };
        float s = 0.0f, mag = 0.0f;
        for( int a = 0; a < 3; ++a )
        {
            const float u = nn[a] * lo[a], v = nn[a] * hi[a];
            s += u > v ? u : v;
            mag += ( u < 0.f ? -u : u ) + ( v < 0.f ? -v : v );
        }
        if( s < -1e-5f * mag ) /* the vertex furthest inside is outside, by more than rounding */
            return false;
    }
    return true;
}
#define VRC_TILE_CANDIDATES 1024u /* per wave; a tile that could hit more runs the plain loop */

/* glRaycaster with nSamplesPerPixel > 1 (fragRaycast.glsl:113-215; one draw per brick, in the host's order): for
 * every brick the fragment casts n rays through gl_FragCoord + (rand, rand) / 2, each marches THIS brick starting
 * from the pixel's colour so far, and the pixel becomes the average of the n results (:212-214).  A `discard` inside
 * the loop (a sub-ray that misses the volume or the brick, :140-143, :159-160, :176-177) discards the fragment: the
 * brick then leaves the pixel as it was.  Sub-sample 0 has rand(0,0) = 0: the pixel centre. */
template < bool CLAMP, bool COUNT, bool FIXED, int MODE, typename ATLAS_T, int GROUP = VRC_GROUP, bool BIG = false >
VRC_HD void vrc_pixel_gl_supersampled( const vrc_frame& f, const vrc_dev_node* __restrict__ nodes,
                                       const ATLAS_T* __restrict__ atlas, const vrc_f4* lut,
                                       const vrc_classifier& cls, vrc_f4* __restrict__ pixelBuffer, uint32_t px,
                                       uint32_t py, uint32_t& nSamples, const uint16_t* candidates,
                                       uint32_t nCandidates )
{
    VRC_STRICT_FP
    const uint32_t row = f.rowMap ? f.rowMap[py] : py;
    const uint32_t pixelPos = py * f.width + px;
    const float fx = (float)px + f.pixelOffX, fy = (float)row + f.pixelOffY; /* gl_FragCoord.xy */
    const vrc_f4 zero = { 0.f, 0.f, 0.f, 0.f };
    vrc_f4 color = f.clearFirst ? zero : pixelBuffer[pixelPos];
    const uint32_t n = f.samplesPerPixel;
    const float fn = (float)n;
    const uint32_t nLoop = candidates ? nCandidates : f.nodeCount;
    for( uint32_t c = 0; c < nLoop; ++c )
    {
        if( color.w > VRC_EARLY_EXIT ) /* :115-117: this and every later brick's fragment is discarded */
            break;
        const uint32_t i = candidates ? (uint32_t)candidates[c] : c;
        const vrc_dev_node node = nodes[i];
        vrc_f4 sum = zero;
        bool discard = false;
        uint32_t cnt = 0;
        for( uint32_t k = 0; k < n; ++k )
        {
            const float fk = (float)k;
            const float dx = vrc_gl_rand( fx * fk, fy * fk ) / 2.0f;
            const float dy = vrc_gl_rand( fx * 2.0f * fk, fy * 2.0f * fk ) / 2.0f;
            const vrc_ray r = vrc_setup_ray_at( f, fx + dx, fy + dy );
            vrc_segment s;
            bool stop;
            if( !r.hit || !vrc_brick_segment( f, r, node, f.stepSize, &s, &stop ) )
            {
                discard = true;
                break;
            }
            vrc_f4 local = color;
            (void)vrc_march_brick< CLAMP, COUNT, FIXED, MODE, ATLAS_T, GROUP, BIG >( f, node, s, atlas, lut, cls, local, cnt );
            sum.x += local.x;
            sum.y += local.y;
            sum.z += local.z;
            sum.w += local.w;
        }
        if( !discard )
        {
            color.x = sum.x / fn;
            color.y = sum.y / fn;
            color.z = sum.z / fn;
            color.w = sum.w / fn;
            if( COUNT )
                nSamples += cnt;
        }
    }
    pixelBuffer[pixelPos] = color;
}

template < bool CLAMP, bool COUNT, bool FIXED, int MODE, typename ATLAS_T, int GROUP = VRC_GROUP, bool BIG = false >
VRC_HD void vrc_pixel_reference_order( const vrc_frame& f, const vrc_dev_node* __restrict__ nodes,
                                       const ATLAS_T* __restrict__ atlas, const vrc_f4* lut,
                                       const vrc_classifier& cls,
                                       vrc_f4* __restrict__ pixelBuffer, uint32_t px, uint32_t py,
                                       uint32_t& nSamples, const uint16_t* candidates = nullptr,
                                       uint32_t nCandidates = 0 )
{
    if( f.variant == VRC_VARIANT_GL && f.samplesPerPixel > 1u )
    {
        vrc_pixel_gl_supersampled< CLAMP, COUNT, FIXED, MODE, ATLAS_T, GROUP, BIG >( f, nodes, atlas, lut, cls, pixelBuffer,
                                                                                   px, py, nSamples, candidates, nCandidates );
        return;
    }
    const vrc_ray r = vrc_setup_ray( f, px, f.rowMap ? f.rowMap[py] : py );
    const uint32_t pixelPos = py * f.width + px;
    const vrc_f4 zero = { 0.f, 0.f, 0.f, 0.f };
    if( !r.hit )
    {
        if( f.clearFirst )
            pixelBuffer[pixelPos] = zero;
        return; /* Renderer.cu:129-130, :148-149: pixel left untouched (= cleared) */
    }
    vrc_f4 color = f.clearFirst ? zero : pixelBuffer[pixelPos];
    if( color.w > VRC_EARLY_EXIT ) /* Renderer.cu:152-155 */
        return;
    /* candidates: the bricks of the list a ray of this wave's tile can hit, in list order (tile culling, above) */
    const uint32_t nLoop = candidates ? nCandidates : f.nodeCount;
    for( uint32_t c = 0; c < nLoop; ++c )
    {
        const uint32_t i = candidates ? (uint32_t)candidates[c] : c;
        const vrc_dev_node n = nodes[i];
        vrc_segment s;
        bool stop;
        if( !vrc_brick_segment( f, r, n, f.stepSize, &s, &stop ) )
        {
            if( stop )
                break;
            continue;
        }
        if( vrc_march_brick< CLAMP, COUNT, FIXED, MODE, ATLAS_T, GROUP, BIG >( f, n, s, atlas, lut, cls, color,
                                                                     nSamples ) )
            break;
    }
    pixelBuffer[pixelPos] = color; /* Renderer.cu:229 */
}

/* ------------------------------------------------------------------------------------------
 * Grid-DDA pixel: same per-brick arithmetic, but the bricks a ray meets are enumerated by a
 * 3-D DDA over the brick grid (cell -> node index table) instead of testing every node.
 * For a regular single-LOD grid the along-ray order equals the reference's host order for
 * every pair of bricks that share a ray ("brick order").
 * ---------------------------------------------------------------------------------------- */
/* the axis the view looks along most, and which way: the slabs of vrc_ray_grid_dda's parts */
VRC_HD int vrc_part_dir( const vrc_frame& f )
{
    const vrc_ray c = vrc_setup_ray( f, f.width / 2u, (uint32_t)( f.vpH * 0.5f ) );
    const float ax = c.dir.x < 0.f ? -c.dir.x : c.dir.x, ay = c.dir.y < 0.f ? -c.dir.y : c.dir.y,
                az = c.dir.z < 0.f ? -c.dir.z : c.dir.z;
    const int a = ax >= ay && ax >= az ? 0 : ( ay >= az ? 1 : 2 );
    const float d = a == 0 ? c.dir.x : ( a == 1 ? c.dir.y : c.dir.z );
    return a | ( d < 0.f ? 4 : 0 );
}

/* The walk of one prepared ray; color is read-modify-write.  part: -1 = every brick the ray meets; 0 .. parts-1 =
 * only the bricks met in that one of `parts` equal slabs of the brick grid along axis partDir & 3, counted in the
 * direction the view looks along it (partDir & 4: towards smaller coordinates; vrc_part_dir).  The slab is a
 * property of the grid cell, the same for every ray: the lanes of a wave agree on which launch marches a brick.
 * partDir < 0 (two parts): the halves of the ray's own interval inside the grid instead.
 * Every brick is marched whole in exactly one part, and a brick's part never falls below that of a brick met
 * before it, so marching the parts one after the other composites the reference's samples in the reference's
 * order.  Two users: the depth split (two waves march the two halves of a tile's rays at the same time and the
 * halves are composited with `over`) and ray compaction (VRC_OPT_ERT_COMPACTION: one launch per part, the rays
 * that early termination has not ended are packed into full waves for the next).
 * Returns whether the ray is still alive (it meets the grid and its opacity is below the early-exit threshold). */
template < bool CLAMP, bool COUNT, bool FIXED, int MODE, typename ATLAS_T, int GROUP = VRC_GROUP, bool BIG = false >
VRC_HD bool vrc_ray_grid_dda( const vrc_frame& f, const vrc_ray& r, const vrc_dev_node* __restrict__ nodes,
                              const int32_t* __restrict__ gridTable, const ATLAS_T* __restrict__ atlas,
                              const vrc_f4* lut, const vrc_classifier& cls, vrc_f4& color, uint32_t& nSamples,
                              int part = -1, int parts = 2, int partDir = 2 )
{
    /* ray interval inside the brick grid */
    const vrc_f3 gmin = { f.gridMin[0], f.gridMin[1], f.gridMin[2] };
    const vrc_f3 gmax = { f.gridMin[0] + f.cellSize[0] * (float)f.gridDim[0],
                          f.gridMin[1] + f.cellSize[1] * (float)f.gridDim[1],
                          f.gridMin[2] + f.cellSize[2] * (float)f.gridDim[2] };
    float t0, t1;
    bool any = vrc_intersect_box( r.origin, r.invDir, gmin, gmax, &t0, &t1 );
    /* the GLSL twin does not clamp a brick's interval to the global box (fragRaycast.glsl:149-150):
     * bricks of the tree that reach past the volume (ragged trees) are sampled there too */
    if( f.variant == VRC_VARIANT_GL )
        t0 = fmaxf( t0, fmaxf( r.tNearPlane, 0.0f ) );
    else
    {
        t0 = fmaxf( fmaxf( t0, r.tNearGlobal ), fmaxf( r.tNearPlane, 0.0f ) );
        t1 = fminf( t1, r.tFarGlobal );
    }
    if( any && t0 <= t1 )
    {
        const float o[3] = { r.origin.x, r.origin.y, r.origin.z };
        const float d[3] = { r.dir.x, r.dir.y, r.dir.z };
        const float id[3] = { r.invDir.x, r.invDir.y, r.invDir.z };
        /* Which bricks does the ray meet?  Those of the cells it runs through -- and, where it enters or leaves
         * a cell through an edge or a corner of the grid (two or three of the entry / exit parameters equal
         * to within rounding), the bricks of the cells around that edge or corner.  In exact arithmetic the
         * ray has no extent in those; the reference tests every brick with its float slab test
         * (Renderer.cu:56-80, :179-181), and one that comes out with tfar a last bit above tnear gets its one
         * sample (:208).  Handing the same cells to the same arithmetic (vrc_brick_segment) reproduces that,
         * instead of depending on which of the tied faces this walk happens to cross first: the walk
         * composites the reference's samples, one for one. */
        int cell[3], stepDir[3];
        float tMax[3], tDelta[3];
        const float tolE = fabsf( t0 ) * 2e-6f;
        uint32_t back = 0u; /* axes on whose cell faces the ray is at t0: it also touches the cells behind them */
#pragma unroll
        for( int a = 0; a < 3; ++a )
        {
            const float p = o[a] + d[a] * t0;
            const float u = ( p - f.gridMin[a] ) * f.invCellSize[a];
            const bool pos = d[a] > 0.0f;
            stepDir[a] = pos ? 1 : -1;
            /* on a cell face at t0 (the ray enters the grid through an edge, or starts on a face): the walk
             * starts in the cell the ray goes on into, so that no cell is left at the parameter it was
             * entered at */
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
        /* bricks already handed to the slab test: a brick is convex, once left it is never entered again,
         * but a coarse brick that spans several cells is met in each of them */
        int32_t recent[4] = { -1, -1, -1, -1 };
        const int maxSteps = f.gridDim[0] + f.gridDim[1] + f.gridDim[2] + 3;
        bool finished = false;
        int lastPart = 0;
        auto visit = [&]( int cx, int cy, int cz ) {
            const int32_t node = gridTable[( cz * f.gridDim[1] + cy ) * f.gridDim[0] + cx];
            if( node < 0 || node == recent[0] || node == recent[1] || node == recent[2] || node == recent[3] )
                return;
            recent[3] = recent[2];
            recent[2] = recent[1];
            recent[1] = recent[0];
            recent[0] = node;
            const vrc_dev_node n = nodes[node];
            vrc_segment s;
            bool stop;
            if( vrc_brick_segment( f, r, n, f.stepSize, &s, &stop ) )
            {
                if( part >= 0 )
                {
                    int sp;
                    if( partDir < 0 )
                        /* by ray parameter: the brick's segment starts in the near / far half of THIS ray's interval
                         * inside the grid -- halves of equal length for every ray (what the depth split wants: the
                         * longer half is the latency), at the price that the lanes of a wave disagree about the
                         * brick the middle falls into */
                        sp = s.tNear < 0.5f * ( t0 + t1 ) ? 0 : 1;
                    else
                    {
                        const int ax = partDir & 3;
                        const int dim = f.gridDim[ax];
                        const int c = ax == 0 ? cx : ( ax == 1 ? cy : cz );
                        sp = ( ( ( partDir & 4 ) ? dim - 1 - c : c ) * parts ) / dim;
                    }
                    sp = sp > lastPart ? sp : lastPart;
                    lastPart = sp;
                    if( sp > part )
                        finished = true; /* nothing of this part lies behind a brick of a later one */
                    if( sp != part )
                        return; /* another part's brick */
                }
                if( vrc_march_brick< CLAMP, COUNT, FIXED, MODE, ATLAS_T, GROUP, BIG >( f, n, s, atlas, lut, cls, color,
                                                                             nSamples ) )
                    finished = true;
            }
            else if( stop )
                finished = true;
        };
        /* work list of the cells around an edge / corner, one bit per cell, in the order they are met:
         *   bits 0-6   cell - (subset of the entry-tied axes): subsets 7, 3, 5, 6, 1, 2, 4
         *   bits 8-13  cell + (proper subset of the exit-tied axes): subsets 1, 2, 4, 3, 5, 6
         * the 64-bit constants hold, per 3-bit axis set, the bits whose subset lies inside it */
        uint32_t pending = (uint32_t)( ( 0x7F68544032201000ull >> ( back * 8u ) ) & 0x7Fu );
        uint32_t tied = 0u; /* axes whose faces the ray leaves the current cell through */
        bool multi = false; /* more than one of them */
        for( int it = 0; it < maxSteps; ++it )
        {
            /* rare, and taken by the whole wave or not at all: the cells around the edge / corner the ray
             * entered the first cell through (first iteration) or left the last cell through */
            bool rare = pending != 0u || multi;
#if defined( __HIP_DEVICE_COMPILE__ )
            rare = __builtin_amdgcn_ballot_w64( rare ) != 0ull;
#endif
            if( rare )
            {
                if( multi )
                    pending |= (uint32_t)( ( 0x3F06050003000000ull >> ( tied * 8u ) ) & 0x3Fu ) << 8;
                while( pending != 0u && !finished )
                {
#if defined( __HIP_DEVICE_COMPILE__ )
                    const uint32_t slot = (uint32_t)__builtin_ctz( pending );
#else
                    uint32_t slot = 0;
                    while( !( ( pending >> slot ) & 1u ) )
                        ++slot;
#endif
                    pending &= pending - 1u;
                    const uint32_t sub = slot < 7u ? ( 0x4216537u >> ( slot * 4u ) ) & 7u
                                                   : ( 0x653421u >> ( ( slot - 8u ) * 4u ) ) & 7u;
                    const int sgn = slot < 7u ? -1 : 1;
                    const int cx = cell[0] + ( ( sub & 1u ) ? sgn * stepDir[0] : 0 );
                    const int cy = cell[1] + ( ( sub & 2u ) ? sgn * stepDir[1] : 0 );
                    const int cz = cell[2] + ( ( sub & 4u ) ? sgn * stepDir[2] : 0 );
                    if( cx >= 0 && cx < f.gridDim[0] && cy >= 0 && cy < f.gridDim[1] && cz >= 0 && cz < f.gridDim[2] )
                        visit( cx, cy, cz );
                }
                pending = 0u;
            }
            if( finished )
                break;
            /* through every tied face at once */
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
            if( cell[0] < 0 || cell[0] >= f.gridDim[0] || cell[1] < 0 || cell[1] >= f.gridDim[1] ||
                cell[2] < 0 || cell[2] >= f.gridDim[2] )
                break;
            visit( cell[0], cell[1], cell[2] );
            if( finished )
                break;
            const float tNext = fminf( fminf( tMax[0], tMax[1] ), tMax[2] );
            if( tNext > t1 ) /* the ray ends inside this cell */
                break;
            const float thr = tNext + fabsf( tNext ) * 2e-6f;
            tied = ( tMax[0] <= thr ? 1u : 0u ) | ( tMax[1] <= thr ? 2u : 0u ) | ( tMax[2] <= thr ? 4u : 0u );
            multi = ( tied & ( tied - 1u ) ) != 0u;
        }
        return !( color.w > VRC_EARLY_EXIT );
    }
    return false;
}

template < bool CLAMP, bool COUNT, bool FIXED, int MODE, typename ATLAS_T, int GROUP = VRC_GROUP, bool BIG = false >
VRC_HD void vrc_pixel_grid_dda( const vrc_frame& f, const vrc_dev_node* __restrict__ nodes,
                                const int32_t* __restrict__ gridTable,
                                const ATLAS_T* __restrict__ atlas, const vrc_f4* lut,
                                const vrc_classifier& cls,
                                vrc_f4* __restrict__ pixelBuffer, uint32_t px, uint32_t py,
                                uint32_t& nSamples )
{
    const vrc_ray r = vrc_setup_ray( f, px, f.rowMap ? f.rowMap[py] : py );
    const uint32_t pixelPos = py * f.width + px;
    const vrc_f4 zero = { 0.f, 0.f, 0.f, 0.f };
    if( !r.hit )
    {
        if( f.clearFirst )
            pixelBuffer[pixelPos] = zero;
        return;
    }
    vrc_f4 color = f.clearFirst ? zero : pixelBuffer[pixelPos];
    if( color.w > VRC_EARLY_EXIT )
        return;
    vrc_ray_grid_dda< CLAMP, COUNT, FIXED, MODE, ATLAS_T, GROUP, BIG >( f, r, nodes, gridTable, atlas, lut, cls, color,
                                                                   nSamples );
    pixelBuffer[pixelPos] = color;
}

/* ------------------------------------------------------------------------------------------
 * EXTENSION: per-ray adaptive LOD (BASELINE C5; vrc_set_ray_lod).  The reference selects the LOD
 * per brick on the host: a brick is fine enough when its voxels, seen from the point of its box
 * nearest to the near plane, are at most screenSpaceError pixels wide
 * (livre/core/render/SelectVisibles.cpp:52-68: pixelPerVoxel * near / (near + distance) <= sse).
 * Here the same criterion is evaluated along the ray.  The node list is a hierarchy (a cut of the
 * octree plus ancestors, all resident, boxes may nest); with voxel size vw0 * 2^j at level j and
 * eye-space depth / near = t / tNearPlane along a ray, level j is fine enough from
 *     T_j = tNearPlane * lodBase * 2^j,   lodBase = vw0 / (screenSpaceError * worldSpacePerPixel).
 * Every level is a regular grid of bricks (border bricks may be smaller; the levels of a ragged
 * tree need not align with each other).  The ray hops from brick to brick: at parameter te it
 * wants level k = #{ j >= 1 : T_j <= te } and takes, at the point o + d*(te + eps), the brick of
 * the first level that has one there in the order k, k+1, ..., K-1, k-1, ..., 0; the run ends
 * where the ray leaves that brick's box (the finest level's cell where there is no brick) and
 * the level is chosen anew.  A run is marched like a reference brick segment
 * (Renderer.cu:195-223: sampling restarts at the run's entry point, here eps inside the brick so
 * that the first sample does not sit on a voxel face) with step stepSize * 2^j and
 * opacity exponent alphaCorrection * 2^j (classified table of level j: lut + j * 257; classifier
 * exponent scaled for the per-sample modes), so a level-j brick costs 2^-j of the samples and
 * the opacity of a homogeneous stretch does not depend on the level it is sampled at.
 * gridTable holds the cell -> node table of every level (f.lodTable offsets).  CUDA variant only.
 * ---------------------------------------------------------------------------------------- */

VRC_HD vrc_segment vrc_run_segment( const vrc_ray& r, float tA, float tB, float stepSize )
{
    VRC_STRICT_FP
    /* the tail of vrc_brick_segment (Renderer.cu:195-201) for an explicit interval */
    const vrc_f3 rayStart = { r.origin.x + r.dir.x * tA, r.origin.y + r.dir.y * tA,
                              r.origin.z + r.dir.z * tA };
    const vrc_f3 rayStop = { r.origin.x + r.dir.x * tB, r.origin.y + r.dir.y * tB,
                             r.origin.z + r.dir.z * tB };
    const vrc_f3 diff = { rayStop.x - rayStart.x, rayStop.y - rayStart.y, rayStop.z - rayStart.z };
    const float d2 = vrc_dot( diff, diff );
    vrc_segment s;
    s.pos = rayStart;
    s.tNear = tA;
    if( d2 > 0.0f )
    {
        const float invLen = 1.0f / sqrtf( d2 );
        s.step.x = diff.x * invLen * stepSize;
        s.step.y = diff.y * invLen * stepSize;
        s.step.z = diff.z * invLen * stepSize;
        s.dist = sqrtf( d2 );
    }
    else
    {
        s.step.x = s.step.y = s.step.z = 0.0f;
        s.dist = 0.0f;
    }
    return s;
}

/* T_0 of the ray: level j is fine enough from T_0 * 2^j on */
VRC_HD float vrc_ray_lod_base( const vrc_frame& f, const vrc_ray& r )
{
    VRC_STRICT_FP
    return r.tNearPlane * f.lodBase;
}

/* One hop of the per-ray LOD walk at ray parameter te: the brick the ray is in at te + eps (its index in
 * `nodes`, copied to n; -1: no level has a brick there), the parameter tp of the run's first sample and tB
 * where the ray leaves the brick's box (no brick: the finest level's grid cell).  Shared by
 * vrc_pixel_ray_lod and the LDS-staged kernel (vrc_kernels_lds.hip): same bits in both. */
VRC_HD int32_t vrc_ray_lod_hop( const vrc_frame& f, const vrc_ray& r, const vrc_dev_node* __restrict__ nodes,
                                const int32_t* __restrict__ gridTable, float tBase, float te, float t1,
                                vrc_dev_node& n, float& tp, float& tB )
{
    const int K = (int)f.lodLevels;
    const float o[3] = { r.origin.x, r.origin.y, r.origin.z };
    const float d[3] = { r.dir.x, r.dir.y, r.dir.z };
    const float id[3] = { r.invDir.x, r.invDir.y, r.invDir.z };
    int k = 0;
    float T = tBase;
    for( int j = 1; j < K; ++j )
    {
        T = T + T; /* T_j, exact */
        k += T <= te ? 1 : 0;
    }
    float p[3];
    {
        VRC_STRICT_FP
        tp = te + f.lodEps;
        p[0] = o[0] + d[0] * tp;
        p[1] = o[1] + d[1] * tp;
        p[2] = o[2] + d[2] * tp;
    }
    int32_t node = -1;
    for( int s = 0; s < K && node < 0; ++s )
    {
        const int lv = s < K - k ? k + s : K - 1 - s; /* k..K-1, then k-1..0 */
        int c[3];
#pragma unroll
        for( int a = 0; a < 3; ++a )
        {
            VRC_STRICT_FP
            c[a] = (int)floorf( ( p[a] - f.gridMin[a] ) * f.lodInvCell[lv][a] );
            c[a] = c[a] < 0 ? 0 : ( c[a] > f.lodDim[lv][a] - 1 ? f.lodDim[lv][a] - 1 : c[a] );
        }
        node = gridTable[f.lodTable[lv] + ( c[2] * f.lodDim[lv][1] + c[1] ) * f.lodDim[lv][0] + c[0]];
    }
    /* where the ray leaves the brick (no brick here: the finest level's grid cell) */
    float bmin[3], bmax[3];
    if( node >= 0 )
    {
        n = nodes[node];
#pragma unroll
        for( int a = 0; a < 3; ++a )
        {
            bmin[a] = n.aabbMin[a];
            bmax[a] = n.aabbMin[a] + n.aabbSize[a];
        }
    }
    else
    {
#pragma unroll
        for( int a = 0; a < 3; ++a )
        {
            VRC_STRICT_FP
            int c = (int)floorf( ( p[a] - f.gridMin[a] ) * f.invCellSize[a] );
            c = c < 0 ? 0 : ( c > f.gridDim[a] - 1 ? f.gridDim[a] - 1 : c );
            bmin[a] = f.gridMin[a] + f.cellSize[a] * (float)c;
            bmax[a] = f.gridMin[a] + f.cellSize[a] * (float)( c + 1 );
        }
    }
    {
        VRC_STRICT_FP
        float tX = ( ( d[0] > 0.0f ? bmax[0] : bmin[0] ) - o[0] ) * id[0];
        tX = fminf( tX, ( ( d[1] > 0.0f ? bmax[1] : bmin[1] ) - o[1] ) * id[1] );
        tX = fminf( tX, ( ( d[2] > 0.0f ? bmax[2] : bmin[2] ) - o[2] ) * id[2] );
        tB = fminf( fmaxf( tX, tp ), t1 ); /* always forward */
    }
    return node;
}

/* the interval of the hierarchy's box a ray crosses (false: none) */
VRC_HD bool vrc_ray_lod_interval( const vrc_frame& f, const vrc_ray& r, float& t0, float& t1 )
{
    const vrc_f3 gmin = { f.gridMin[0], f.gridMin[1], f.gridMin[2] };
    const vrc_f3 gmax = { f.lodMax[0], f.lodMax[1], f.lodMax[2] };
    const bool any = vrc_intersect_box( r.origin, r.invDir, gmin, gmax, &t0, &t1 );
    t0 = fmaxf( fmaxf( t0, r.tNearGlobal ), fmaxf( r.tNearPlane, 0.0f ) );
    t1 = fminf( t1, r.tFarGlobal );
    return any && t0 < t1;
}
VRC_HD int vrc_ray_lod_max_hops( const vrc_frame& f )
{
    return 3 * ( f.gridDim[0] + f.gridDim[1] + f.gridDim[2] ) + 16;
}

template < bool CLAMP, bool COUNT, bool FIXED, int MODE, typename ATLAS_T, int GROUP = VRC_GROUP, bool BIG = false >
VRC_HD void vrc_pixel_ray_lod( const vrc_frame& f, const vrc_dev_node* __restrict__ nodes,
                               const int32_t* __restrict__ gridTable,
                               const ATLAS_T* __restrict__ atlas, const vrc_f4* lut,
                               const vrc_classifier& cls,
                               vrc_f4* __restrict__ pixelBuffer, uint32_t px, uint32_t py,
                               uint32_t& nSamples )
{
    const vrc_ray r = vrc_setup_ray( f, px, f.rowMap ? f.rowMap[py] : py );
    const uint32_t pixelPos = py * f.width + px;
    const vrc_f4 zero = { 0.f, 0.f, 0.f, 0.f };
    if( !r.hit )
    {
        if( f.clearFirst )
            pixelBuffer[pixelPos] = zero;
        return;
    }
    vrc_f4 color = f.clearFirst ? zero : pixelBuffer[pixelPos];
    if( color.w > VRC_EARLY_EXIT )
        return;

    float t0, t1;
    if( vrc_ray_lod_interval( f, r, t0, t1 ) )
    {
        const float tBase = vrc_ray_lod_base( f, r );
        const int maxHops = vrc_ray_lod_max_hops( f );
        float te = t0;
        for( int hop = 0; hop < maxHops && te < t1; ++hop )
        {
            vrc_dev_node n;
            float tp, tB;
            const int32_t node = vrc_ray_lod_hop( f, r, nodes, gridTable, tBase, te, t1, n, tp, tB );
            if( node >= 0 )
            {
                const float scale = (float)( 1u << n.level );
                const float levelStep = f.stepSize * scale;
                const vrc_segment s = vrc_run_segment( r, tp, tB, levelStep ); /* first sample eps inside */
                vrc_classifier lc = cls;
                lc.alphaCorrection = cls.alphaCorrection * scale;
                /* the level's classified table (257 entries of four floats, or of two in the grey form) */
                const vrc_f4* ll = MODE == VRC_MODE_TABLE ? lut + n.level * VRC_LUT_ENTRIES
                                   : MODE == VRC_MODE_GREY
                                       ? reinterpret_cast< const vrc_f4* >( reinterpret_cast< const vrc_f2* >( lut ) +
                                                                            n.level * VRC_LUT_ENTRIES )
                                       : lut;
                if( vrc_march_brick< CLAMP, COUNT, FIXED, MODE, ATLAS_T, GROUP, BIG >( f, n, s, atlas, ll, lc, color,
                                                                             nSamples, levelStep ) )
                    break;
            }
            te = tB;
        }
    }
    pixelBuffer[pixelPos] = color;
}

#endif /* VRC_CORE_H */

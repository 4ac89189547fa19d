/*
 * livre_oracle.c -- CPU ORACLE (test infrastructure, see livre_oracle.h).
 *
 * Plain C, float32 arithmetic with no FMA contraction (build with -ffp-contract=off),
 * restating the reference algorithm function by function.  Citations are path:line
 * relative to the reference root.  "parity unpinned" for pixels (the reference has no
 * golden frame); host-side pieces are pinned by the reference's unit-test known answers.
 */
#include "livre_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------
 * NodeId -- livre/core/data/NodeId.h:38-49: bit-field, LSB first:
 * level:4 | x:14 | y:14 | z:14 | timeStep:18   (types.h:191-192, mathTypes.h:82)
 * ---------------------------------------------------------------------------------- */
#define LEVEL_BITS 4u
#define BLOCK_BITS 14u
#define TIME_BITS 18u
#define INVALID_LEVEL ( ( 1u << LEVEL_BITS ) - 1u ) /* types.h:195 */

uint64_t orc_nodeid_pack( uint32_t level, uint32_t x, uint32_t y, uint32_t z, uint32_t t )
{
    uint64_t id = 0;
    id |= (uint64_t)( level & 0xFu );
    id |= (uint64_t)( x & 0x3FFFu ) << 4;
    id |= (uint64_t)( y & 0x3FFFu ) << 18;
    id |= (uint64_t)( z & 0x3FFFu ) << 32;
    id |= (uint64_t)( t & 0x3FFFFu ) << 46;
    return id;
}

void orc_nodeid_unpack( uint64_t id, uint32_t out[5] )
{
    out[0] = (uint32_t)( id & 0xFu );
    out[1] = (uint32_t)( ( id >> 4 ) & 0x3FFFu );
    out[2] = (uint32_t)( ( id >> 18 ) & 0x3FFFu );
    out[3] = (uint32_t)( ( id >> 32 ) & 0x3FFFu );
    out[4] = (uint32_t)( ( id >> 46 ) & 0x3FFFFu );
}

/* NodeId.cpp:61-68 */
uint64_t orc_nodeid_parent( uint64_t id )
{
    uint32_t f[5];
    orc_nodeid_unpack( id, f );
    if( f[0] == INVALID_LEVEL || f[0] == 0 )
        return ~(uint64_t)0; /* INVALID_NODE_ID */
    return orc_nodeid_pack( f[0] - 1, f[1] / 2, f[2] / 2, f[3] / 2, f[4] );
}

/* NodeId.cpp:92-113: x outer, y middle, z inner */
void orc_nodeid_children( uint64_t id, uint64_t out[8] )
{
    uint32_t f[5];
    orc_nodeid_unpack( id, f );
    int k = 0;
    for( uint32_t x = 0; x < 2; ++x )
        for( uint32_t y = 0; y < 2; ++y )
            for( uint32_t z = 0; z < 2; ++z )
                out[k++] = orc_nodeid_pack( f[0] + 1, f[1] * 2 + x, f[2] * 2 + y,
                                            f[3] * 2 + z, f[4] );
}

/* ------------------------------------------------------------------------------------
 * Volume information -- livre/core/data/DataSourcePlugin.cpp:83-109
 * ---------------------------------------------------------------------------------- */
static uint32_t max3u( const uint32_t v[3] )
{
    uint32_t m = v[0];
    if( v[1] > m ) m = v[1];
    if( v[2] > m ) m = v[2];
    return m;
}

/* vmmlib find_max_index: first index of the maximum */
static int max_index3u( const uint32_t v[3] )
{
    int idx = 0;
    if( v[1] > v[idx] ) idx = 1;
    if( v[2] > v[idx] ) idx = 2;
    return idx;
}

void orc_fill_regular_volume_info( orc_volume_info* info )
{
    info->worldSpacePerVoxel = 1.0f / (float)max3u( info->voxels );
    for( int i = 0; i < 3; ++i )
        info->worldSize[i] = (float)info->voxels[i] * info->worldSpacePerVoxel;

    uint32_t blockSize[3], numBlocks[3], lodLevels[3];
    for( int i = 0; i < 3; ++i )
    {
        blockSize[i] = info->maximumBlockSize[i] - info->overlap[i] * 2;
        numBlocks[i] = (uint32_t)ceilf( (float)info->voxels[i] / (float)blockSize[i] );
        lodLevels[i] = (uint32_t)ceil( log2( (double)numBlocks[i] ) );
    }
    uint32_t depth = lodLevels[0];
    if( lodLevels[1] < depth ) depth = lodLevels[1];
    if( lodLevels[2] < depth ) depth = lodLevels[2];
    for( int i = 0; i < 3; ++i )
        info->rootBlocks[i] =
            (uint32_t)ceilf( (float)( info->voxels[i] >> depth ) / (float)blockSize[i] );
    info->depth = depth + 1;
}

/* datasources/memory/MemoryDataSource.cpp:74-131: overlap 4, block + 2*overlap */
void orc_mem_volume_info( uint32_t vx, uint32_t vy, uint32_t vz, uint32_t block,
                          orc_volume_info* info )
{
    memset( info, 0, sizeof( *info ) );
    info->voxels[0] = vx;
    info->voxels[1] = vy;
    info->voxels[2] = vz;
    for( int i = 0; i < 3; ++i )
    {
        info->overlap[i] = 4;
        info->maximumBlockSize[i] = block + 8;
    }
    orc_fill_regular_volume_info( info );
}

/* DataSourcePlugin.cpp:55-81 + LODNode.cpp:62-66 */
void orc_lod_node_from_id( const orc_volume_info* info, uint64_t nodeId, orc_lod_node* out )
{
    uint32_t f[5];
    orc_nodeid_unpack( nodeId, f );
    const uint32_t level = f[0];
    uint32_t bricksInRefLevel[3];
    for( int i = 0; i < 3; ++i )
        bricksInRefLevel[i] = info->rootBlocks[i] * ( 1u << level ); /* NodeId.h:162-163 */
    const int index = max_index3u( bricksInRefLevel );
    const float denom = (float)bricksInRefLevel[index];

    out->nodeId = nodeId;
    for( int i = 0; i < 3; ++i )
    {
        const float cmin = (float)f[1 + i] / denom;
        const float cmax = (float)( f[1 + i] + 1u ) / denom;
        const float half = info->worldSize[i] * 0.5f;
        out->worldBoxMin[i] = cmin - half;
        out->worldBoxMax[i] = cmax - half;
        out->blockSize[i] = info->maximumBlockSize[i] - info->overlap[i] * 2;
        out->voxelBoxMin[i] = f[1 + i] * out->blockSize[i];
        out->voxelBoxMax[i] = out->voxelBoxMin[i] + out->blockSize[i];
    }
}

/* MemoryDataSource.cpp:54-57: (id0^id1^id2^id3) + 16 + 127*sin((t+1)/200), T=uint8 */
uint8_t orc_mem_brick_value_u8( uint64_t nodeId )
{
    const uint8_t b0 = (uint8_t)( nodeId & 0xFF );
    const uint8_t b1 = (uint8_t)( ( nodeId >> 8 ) & 0xFF );
    const uint8_t b2 = (uint8_t)( ( nodeId >> 16 ) & 0xFF );
    const uint8_t b3 = (uint8_t)( ( nodeId >> 24 ) & 0xFF );
    uint32_t f[5];
    orc_nodeid_unpack( nodeId, f );
    /* int + int + (int * float): the sum is evaluated in float, then converted to T */
    const float value = (float)( ( b0 ^ b1 ^ b2 ^ b3 ) + 16 ) +
                        127.0f * sinf( ( (float)f[4] + 1.0f ) / 200.f );
    return (uint8_t)value;
}

void orc_mem_brick_fill_u8( const orc_volume_info* info, uint64_t nodeId, uint8_t* dst )
{
    const size_t n = (size_t)info->maximumBlockSize[0] * info->maximumBlockSize[1] *
                     info->maximumBlockSize[2];
    memset( dst, orc_mem_brick_value_u8( nodeId ), n );
}

/* ------------------------------------------------------------------------------------
 * Texture pool -- renderers/cudaRaycaster/cuda/TexturePool.cu
 * ---------------------------------------------------------------------------------- */
static uint32_t minu( uint32_t a, uint32_t b ) { return a < b ? a : b; }
static uint32_t maxu( uint32_t a, uint32_t b ) { return a > b ? a : b; }

/* TexturePool.cu:122-135.  (The reference truncates maxMemory to uint32_t, quirk Q12;
 * the oracle keeps 64 bits as the product does.) */
void orc_pool_slots( const uint32_t maxBlock[3], size_t slotBytes, size_t maxBytes,
                     const uint32_t maxTexture3D[3], uint32_t s[3] )
{
    const size_t maxBlocks64 = maxBytes / slotBytes;
    const uint32_t maxBlocks = maxBlocks64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)maxBlocks64;
    s[0] = minu( maxTexture3D[0] / maxBlock[0], maxu( maxBlocks, 1u ) );
    s[1] = minu( maxTexture3D[1] / maxBlock[1], maxu( maxBlocks / s[0], 1u ) );
    s[2] = minu( maxTexture3D[2] / maxBlock[2], maxu( maxBlocks / ( s[0] * s[1] ), 1u ) );
}

/* TexturePool.cu:137-144 pushes (i,j,k) for i,j,k descending with k innermost; :183-184
 * pops from the back, so the first slot handed out is (0,0,0), then k ascends first. */
void orc_pool_kth_slot( const uint32_t s[3], uint32_t kth, float slot[3] )
{
    const uint32_t k = kth % s[2];
    const uint32_t j = ( kth / s[2] ) % s[1];
    const uint32_t i = kth / ( s[2] * s[1] );
    slot[0] = (float)i / (float)s[0];
    slot[1] = (float)j / (float)s[1];
    slot[2] = (float)k / (float)s[2];
}

/* TexturePool.cu:193-197 */
void orc_pool_slot_voxel_origin( const uint32_t s[3], const uint32_t maxBlock[3],
                                 const float slot[3], uint32_t origin[3] )
{
    for( int a = 0; a < 3; ++a )
    {
        const float volumeSize = (float)( s[a] * maxBlock[a] );
        origin[a] = (uint32_t)lroundf( slot[a] * volumeSize );
    }
}

void orc_pool_copy_to_slot_u8( uint8_t* atlas, const uint32_t dim[3], const uint32_t origin[3],
                               const uint8_t* src, const uint32_t size[3] )
{
    for( uint32_t z = 0; z < size[2]; ++z )
        for( uint32_t y = 0; y < size[1]; ++y )
        {
            uint8_t* d = atlas + ( (size_t)( origin[2] + z ) * dim[1] + ( origin[1] + y ) ) *
                                     dim[0] + origin[0];
            memcpy( d, src + ( (size_t)z * size[1] + y ) * size[0], size[0] );
        }
}

/* CudaTextureObject.cpp:61-84 */
void orc_texture_object( const orc_volume_info* info, const orc_lod_node* node,
                         const float slot[3], const uint32_t atlasDim[3], float texPos[3],
                         float texSize[3] )
{
    for( int a = 0; a < 3; ++a )
    {
        const float cacheTextureSize = (float)atlasDim[a];
        const float overlapf = (float)info->overlap[a] / cacheTextureSize;
        const float size = (float)( node->voxelBoxMax[a] - node->voxelBoxMin[a] );
        texPos[a] = slot[a] + overlapf;
        texSize[a] = size / cacheTextureSize;
    }
}

/* ------------------------------------------------------------------------------------
 * Matrices (column-major float[16], as vmmlib stores them and cuda/math.cuh:1457 reads)
 * ---------------------------------------------------------------------------------- */
#define M( m, r, c ) ( ( m )[( c ) * 4 + ( r )] )

void orc_mat4_identity( float m[16] )
{
    memset( m, 0, 16 * sizeof( float ) );
    m[0] = m[5] = m[10] = m[15] = 1.0f;
}

void orc_mat4_mul( const float a[16], const float b[16], float out[16] )
{
    float r[16];
    for( int c = 0; c < 4; ++c )
        for( int row = 0; row < 4; ++row )
        {
            float s = 0.f;
            for( int k = 0; k < 4; ++k )
                s += M( a, row, k ) * M( b, k, c );
            M( r, row, c ) = s;
        }
    memcpy( out, r, sizeof( r ) );
}

/* general 4x4 inverse by cofactors (vmmlib Matrix4::inverse, call site Frustum.cpp:31,34);
 * evaluated in double and rounded once, so it is the correctly rounded inverse */
int orc_mat4_inverse( const float mf[16], float out[16] )
{
    double m[16], inv[16];
    for( int i = 0; i < 16; ++i ) m[i] = mf[i];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] +
             m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] -
             m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] +
             m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] -
              m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] -
             m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] +
             m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] -
             m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] +
              m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] +
             m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] -
             m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] +
              m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] -
              m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] -
             m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] +
             m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] -
              m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] +
              m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if( det == 0.0 )
        return 0;
    for( int i = 0; i < 16; ++i )
        out[i] = (float)( inv[i] / det );
    return 1;
}

static void normalize3( float v[3] )
{
    const float l = sqrtf( v[0] * v[0] + v[1] * v[1] + v[2] * v[2] );
    v[0] /= l;
    v[1] /= l;
    v[2] /= l;
}

static void cross3( const float a[3], const float b[3], float o[3] )
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

/* vmmlib Matrix4(eye, lookAt, up) as used at CameraSettings.cpp:102; pinned by the
 * known answer at tests/eq/settings/cameraSettings.cpp:99-117 (gluLookAt convention). */
void orc_look_at( const float eye[3], const float center[3], const float up[3], float out[16] )
{
    float f[3] = { center[0] - eye[0], center[1] - eye[1], center[2] - eye[2] };
    normalize3( f );
    float s[3], u[3];
    cross3( f, up, s );
    normalize3( s );
    cross3( s, f, u );
    orc_mat4_identity( out );
    for( int c = 0; c < 3; ++c )
    {
        M( out, 0, c ) = s[c];
        M( out, 1, c ) = u[c];
        M( out, 2, c ) = -f[c];
    }
    M( out, 0, 3 ) = -( s[0] * eye[0] + s[1] * eye[1] + s[2] * eye[2] );
    M( out, 1, 3 ) = -( u[0] * eye[0] + u[1] * eye[1] + u[2] * eye[2] );
    M( out, 2, 3 ) = f[0] * eye[0] + f[1] * eye[1] + f[2] * eye[2];
}

/* CameraSettings.cpp:35-59; rotation sign convention of vmmlib pre_rotate_x/y pinned by
 * tests/eq/settings/cameraSettings.cpp:44-57 */
void orc_spin_model( float mv[16], float x, float y )
{
    if( x == 0.f && y == 0.f )
        return;
    const float tx = M( mv, 0, 3 ), ty = M( mv, 1, 3 ), tz = M( mv, 2, 3 );
    M( mv, 0, 3 ) = M( mv, 1, 3 ) = M( mv, 2, 3 ) = 0.f;

    float rx[16], ry[16], tmp[16];
    orc_mat4_identity( rx );
    orc_mat4_identity( ry );
    const float cx = cosf( x ), sx = sinf( x ), cy = cosf( y ), sy = sinf( y );
    M( rx, 1, 1 ) = cx;  M( rx, 1, 2 ) = sx;
    M( rx, 2, 1 ) = -sx; M( rx, 2, 2 ) = cx;
    M( ry, 0, 0 ) = cy;  M( ry, 0, 2 ) = -sy;
    M( ry, 2, 0 ) = sy;  M( ry, 2, 2 ) = cy;
    orc_mat4_mul( rx, mv, tmp );
    orc_mat4_mul( ry, tmp, mv );

    M( mv, 0, 3 ) = tx;
    M( mv, 1, 3 ) = ty;
    M( mv, 2, 3 ) = tz;
}

/* glFrustum-style perspective (eq::Frustumf::computePerspectiveMatrix, call site
 * livre/eq/Channel.cpp:154-155); with l/r/b/t = -/+0.05, n = 0.1, f = 15 it reproduces
 * the matrix at tests/lib/lodSelection.cpp:38-41 */
void orc_perspective_frustum( float l, float r, float b, float t, float n, float f,
                              float out[16] )
{
    memset( out, 0, 16 * sizeof( float ) );
    M( out, 0, 0 ) = 2.f * n / ( r - l );
    M( out, 1, 1 ) = 2.f * n / ( t - b );
    M( out, 0, 2 ) = ( r + l ) / ( r - l );
    M( out, 1, 2 ) = ( t + b ) / ( t - b );
    M( out, 2, 2 ) = -( f + n ) / ( f - n );
    M( out, 3, 2 ) = -1.f;
    M( out, 2, 3 ) = -2.f * f * n / ( f - n );
}

/* CudaRaycastRenderer.cpp:136-152 + Frustum.cpp:27-43 (eye = translation of MV^-1) */
void orc_make_view_data( const float mv[16], const float proj[16], const uint32_t viewport[4],
                    const orc_volume_info* info, orc_view_data* out )
{
    memset( out, 0, sizeof( *out ) );
    memcpy( out->modelViewMatrix, mv, 16 * sizeof( float ) );
    orc_mat4_inverse( mv, out->invViewMatrix );
    orc_mat4_inverse( proj, out->invProjMatrix );
    for( int i = 0; i < 3; ++i )
    {
        out->eyePosition[i] = M( out->invViewMatrix, i, 3 );
        out->aabbMin[i] = -( info->worldSize[i] / 2.0f );
        out->aabbMax[i] = info->worldSize[i] / 2.0f;
    }
    for( int i = 0; i < 4; ++i )
        out->glViewport[i] = viewport[i];
    /* vmml::Frustum(proj)::nearPlane(): n = m[14] / (m[10] - 1) for a perspective matrix */
    out->nearPlane = M( proj, 2, 3 ) / ( M( proj, 2, 2 ) - 1.0f );
}

/* CudaRaycastRenderer.cpp:41-61 */
float orc_node_distance( const float mv[16], const orc_lod_node* node )
{
    float c[3];
    for( int i = 0; i < 3; ++i )
        c[i] = ( node->worldBoxMin[i] + node->worldBoxMax[i] ) * 0.5f;
    /* vmmlib Matrix4 * Vector3: homogeneous transform with w = 1 and divide */
    float r[4];
    for( int row = 0; row < 4; ++row )
        r[row] = M( mv, row, 0 ) * c[0] + M( mv, row, 1 ) * c[1] + M( mv, row, 2 ) * c[2] +
                 M( mv, row, 3 );
    const float x = r[0] / r[3], y = r[1] / r[3], z = r[2] / r[3];
    return sqrtf( x * x + y * y + z * z );
}

typedef struct
{
    float d;
    uint64_t id;
    uint32_t index; /* position in the caller's list */
} sort_item;

static int cmp_sort_item( const void* a, const void* b )
{
    const sort_item* x = (const sort_item*)a;
    const sort_item* y = (const sort_item*)b;
    if( x->d < y->d ) return -1;
    if( x->d > y->d ) return 1;
    /* std::sort (CudaRaycastRenderer.cpp:160-163) leaves the order of bricks at equal centre distance
     * to the library; here they keep the order of the caller's list, which makes the oracle
     * deterministic and lets a test hand over the list exactly as a host produced it.  In a regular
     * grid bricks at equal distance share a ray only where it runs along their common face. */
    return x->index < y->index ? -1 : ( x->index > y->index ? 1 : 0 );
}

void orc_sort_nodes_front_to_back( const orc_volume_info* info, const float mv[16],
                                   uint64_t* ids, uint32_t n )
{
    sort_item* items = (sort_item*)malloc( sizeof( sort_item ) * ( n ? n : 1 ) );
    for( uint32_t i = 0; i < n; ++i )
    {
        orc_lod_node node;
        orc_lod_node_from_id( info, ids[i], &node );
        items[i].d = orc_node_distance( mv, &node );
        items[i].id = ids[i];
        items[i].index = i;
    }
    qsort( items, n, sizeof( sort_item ), cmp_sort_item );
    for( uint32_t i = 0; i < n; ++i )
        ids[i] = items[i].id;
    free( items );
}

/* CudaRaycastRenderer.cpp:113-129 (the flag value is used when non-zero, as the GL twin
 * does at GLRaycastRenderer.cpp:226; quirk Q7) */
uint32_t orc_computed_samples_per_ray( const orc_volume_info* info, const uint64_t* ids,
                                       uint32_t n, uint32_t flag )
{
    if( flag != 0 )
        return flag;
    uint32_t maxLOD = 0;
    for( uint32_t i = 0; i < n; ++i )
    {
        const uint32_t level = (uint32_t)( ids[i] & 0xFu );
        if( level > maxLOD )
            maxLOD = level;
    }
    const float maxVoxelDim = (float)max3u( info->voxels );
    const float maxVoxelsAtLOD = maxVoxelDim / (float)( 1u << ( info->depth - maxLOD - 1 ) );
    const float v = maxVoxelsAtLOD > 512.0f ? maxVoxelsAtLOD : 512.0f; /* :65-66, :128 */
    return (uint32_t)v;
}

/* ------------------------------------------------------------------------------------
 * The integrator -- renderers/cudaRaycaster/cuda/Renderer.cu:34-230
 * ---------------------------------------------------------------------------------- */
#define EARLY_EXIT 0.999f          /* Renderer.cu:34 */
#define EPSILON 0.0000000001f      /* Renderer.cu:35 */

typedef struct { float x, y, z; } f3;
typedef struct { float x, y, z, w; } f4;

/* cuda/math.cuh:1457-1464 */
static f4 mat_mul_vec4( const float* m, f4 v )
{
    f4 r;
    r.x = m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * v.w;
    r.y = m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * v.w;
    r.z = m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * v.w;
    r.w = m[3] * v.x + m[7] * v.y + m[11] * v.z + m[15] * v.w;
    return r;
}

static float dot3( f3 a, f3 b ) { return a.x * b.x + a.y * b.y + a.z * b.z; }

/* cuda/math.cuh:1310-1314: v * rsqrtf(dot(v,v)) */
static f3 normalize_f3( f3 v )
{
    const float invLen = 1.0f / sqrtf( dot3( v, v ) );
    f3 r = { v.x * invLen, v.y * invLen, v.z * invLen };
    return r;
}

/* Renderer.cu:40-51 */
static f4 eye_space_from_window( float wx, float wy, const uint32_t vp[4], const float* invProj )
{
    const float vz = (float)vp[2], vw = (float)vp[3];
    const float nx = 2.0f * ( wx - (float)vp[0] - ( vz / 2.0f ) ) / vz;
    const float ny = 2.0f * ( wy - (float)vp[1] - ( vw / 2.0f ) ) / vw;
    const f4 ndc = { nx, ny, 1.0f, 1.0f };
    const f4 e = mat_mul_vec4( invProj, ndc );
    const f4 r = { e.x / e.w, e.y / e.w, e.z / e.w, e.w / e.w };
    return r;
}

/* Renderer.cu:56-80 */
static int intersect_box( f3 origin, f3 dir, f3 boxMin, f3 boxMax, float* tnear, float* tfar )
{
    const f3 invR = { 1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z };
    const f3 tbot = { invR.x * ( boxMin.x - origin.x ), invR.y * ( boxMin.y - origin.y ),
                      invR.z * ( boxMin.z - origin.z ) };
    const f3 ttop = { invR.x * ( boxMax.x - origin.x ), invR.y * ( boxMax.y - origin.y ),
                      invR.z * ( boxMax.z - origin.z ) };
    const f3 tmin = { fminf( ttop.x, tbot.x ), fminf( ttop.y, tbot.y ), fminf( ttop.z, tbot.z ) };
    const f3 tmax = { fmaxf( ttop.x, tbot.x ), fmaxf( ttop.y, tbot.y ), fmaxf( ttop.z, tbot.z ) };
    const float largestTmin = fmaxf( fmaxf( tmin.x, tmin.y ), fmaxf( tmin.x, tmin.z ) );
    const float smallestTmax = fminf( fminf( tmax.x, tmax.y ), fminf( tmax.x, tmax.z ) );
    *tnear = largestTmin;
    *tfar = smallestTmax;
    return smallestTmax > largestTmin;
}

/* Renderer.cu:83-93 */
void orc_composite( const float src[4], float dst[4], float alphaCorrection )
{
    const float corr = 1.0f - fminf( src[3], 1.0f - 1.0f / 256.0f );
    const float alpha = 1.0f - powf( corr, alphaCorrection );
    const float oneMinusDstW = 1.0f - dst[3];
    dst[0] = dst[0] + src[0] * alpha * oneMinusDstW;
    dst[1] = dst[1] + src[1] * alpha * oneMinusDstW;
    dst[2] = dst[2] + src[2] * alpha * oneMinusDstW;
    dst[3] = dst[3] + alpha * oneMinusDstW;
}

/* tex1D<float4>, 256 texels, linear, normalized, clamp (cuda/ColorMap.cu:40-45).
 * CUDA linear filtering: xB = u*N - 0.5, i = floor(xB), a = frac(xB) kept in 1.8 fixed
 * point; result (1-a)*T[i] + a*T[i+1] with clamped indices. */
void orc_tf_fetch( const float* tf, float u, int fracBits, float out[4] )
{
    const float N = 256.0f;
    const float xB = u * N - 0.5f;
    const float fl = floorf( xB );
    float a = xB - fl;
    if( fracBits > 0 )
    {
        const float q = (float)( 1 << fracBits );
        a = floorf( a * q + 0.5f ) / q;
    }
    int i0 = (int)fl, i1 = (int)fl + 1;
    if( i0 < 0 ) i0 = 0;
    if( i0 > 255 ) i0 = 255;
    if( i1 < 0 ) i1 = 0;
    if( i1 > 255 ) i1 = 255;
    for( int c = 0; c < 4; ++c )
        out[c] = ( 1.0f - a ) * tf[i0 * 4 + c] + a * tf[i1 * 4 + c];
}

/* tex3D<uchar>, point filter, normalized coords, clamp, element read mode
 * (cuda/TexturePool.cu:163-170): voxel = clamp(floor(u*N), 0, N-1) per axis */
static int tex_index( float u, uint32_t n )
{
    const float f = floorf( u * (float)n );
    if( !( f > 0.0f ) ) return 0;
    if( f > (float)( n - 1 ) ) return (int)n - 1;
    return (int)f;
}

static float texel( const uint8_t* atlas, int voxelBytes, size_t i )
{
    return voxelBytes == 2 ? (float)( (const uint16_t*)atlas )[i] : (float)atlas[i];
}

static float fetch_nearest( const uint8_t* atlas, int voxelBytes, const uint32_t dim[3], f3 t )
{
    const int x = tex_index( t.x, dim[0] );
    const int y = tex_index( t.y, dim[1] );
    const int z = tex_index( t.z, dim[2] );
    return texel( atlas, voxelBytes, ( (size_t)z * dim[1] + (size_t)y ) * dim[0] + (size_t)x );
}

/* EXTENSION (not in the reference, which is point-sampled): trilinear fetch with exact
 * float weights, texel centres at i + 0.5, clamp addressing.  Returns the raw voxel scale. */
static float fetch_trilinear( const uint8_t* atlas, int voxelBytes, const uint32_t dim[3], f3 t )
{
    const float c[3] = { t.x * (float)dim[0] - 0.5f, t.y * (float)dim[1] - 0.5f,
                         t.z * (float)dim[2] - 0.5f };
    int i0[3], i1[3];
    float w[3];
    for( int a = 0; a < 3; ++a )
    {
        const float fl = floorf( c[a] );
        w[a] = c[a] - fl;
        int lo = (int)fl, hi = (int)fl + 1;
        if( lo < 0 ) lo = 0;
        if( hi < 0 ) hi = 0;
        if( lo > (int)dim[a] - 1 ) lo = (int)dim[a] - 1;
        if( hi > (int)dim[a] - 1 ) hi = (int)dim[a] - 1;
        i0[a] = lo;
        i1[a] = hi;
    }
#define AT( X, Y, Z ) texel( atlas, voxelBytes, ( (size_t)( Z ) * dim[1] + (size_t)( Y ) ) * dim[0] + (size_t)( X ) )
    const float c00 = AT( i0[0], i0[1], i0[2] ) * ( 1.0f - w[0] ) + AT( i1[0], i0[1], i0[2] ) * w[0];
    const float c10 = AT( i0[0], i1[1], i0[2] ) * ( 1.0f - w[0] ) + AT( i1[0], i1[1], i0[2] ) * w[0];
    const float c01 = AT( i0[0], i0[1], i1[2] ) * ( 1.0f - w[0] ) + AT( i1[0], i0[1], i1[2] ) * w[0];
    const float c11 = AT( i0[0], i1[1], i1[2] ) * ( 1.0f - w[0] ) + AT( i1[0], i1[1], i1[2] ) * w[0];
#undef AT
    const float c0 = c00 * ( 1.0f - w[1] ) + c10 * w[1];
    const float c1 = c01 * ( 1.0f - w[1] ) + c11 * w[1];
    return c0 * ( 1.0f - w[2] ) + c1 * w[2];
}

struct lod_grid;
typedef struct
{
    const uint8_t* atlas;
    const uint32_t* atlasDim;
    float* pixelBuffer;
    uint32_t width, height;
    const float* clipPlanes;
    uint32_t nPlanes;
    const float* tf;
    const orc_view_data* view;
    uint32_t nodeCount;
    const orc_node_data* nodes;
    const orc_render_data* render;
    orc_options opt;
    const struct lod_grid* lod; /* rayLod only */
} job_t;

/* TEST INSTRUMENT (orc_options.tieBudget), not part of the restated algorithm: the largest change of a
 * pixel channel if this point sample read the voxel across a face it lies within tieDelta voxels of.
 * transmittance = 1 - accumulated alpha before the sample. */
static float tie_budget( const job_t* j, f3 texPos, float density, float multiplyer, float addedValue,
                         float alphaCorrection, float transmittance, uint32_t k, f3 voxelsPerWorld )
{
    /* how close to a voxel face counts: tieDelta voxels for the coordinate evaluation itself, plus what the
     * reference's own position chain (pos += step, Renderer.cu:208) may have drifted after k additions:
     * half an ulp of a coordinate below 1 (2^-25 world units) per addition, in voxels */
    const float drift = (float)k * 2.98023224e-8f;
    const float delta[3] = { j->opt.tieDelta + drift * voxelsPerWorld.x, j->opt.tieDelta + drift * voxelsPerWorld.y,
                             j->opt.tieDelta + drift * voxelsPerWorld.z };
    const float c[3] = { texPos.x * (float)j->atlasDim[0], texPos.y * (float)j->atlasDim[1],
                         texPos.z * (float)j->atlasDim[2] };
    int idx[3], off[3] = { 0, 0, 0 }, any = 0;
    for( int a = 0; a < 3; ++a )
    {
        const float fl = floorf( c[a] );
        const float fr = c[a] - fl;
        idx[a] = tex_index( a == 0 ? texPos.x : a == 1 ? texPos.y : texPos.z, j->atlasDim[a] );
        if( fr < delta[a] && idx[a] > 0 )
            off[a] = -1;
        else if( fr > 1.0f - delta[a] && idx[a] < (int)j->atlasDim[a] - 1 )
            off[a] = 1;
        any |= off[a] != 0;
    }
    if( !any )
        return 0.0f;
    float tfn[4], here[4] = { 0, 0, 0, 0 }, worst = 0.0f;
    orc_tf_fetch( j->tf, density * multiplyer + addedValue, j->opt.tfFracBits, tfn );
    orc_composite( tfn, here, alphaCorrection ); /* (rgb*alpha', alpha') of this sample */
    /* every combination of the near faces (a sample near an edge or a corner) */
    for( int m = 1; m < 8; ++m )
    {
        int q[3], ok = 1;
        for( int a = 0; a < 3; ++a )
        {
            const int use = ( m >> a ) & 1;
            if( use && off[a] == 0 )
                ok = 0;
            q[a] = idx[a] + ( use ? off[a] : 0 );
        }
        if( !ok )
            continue;
        const float d2 = texel( j->atlas, j->opt.voxelBytes,
                                ( (size_t)q[2] * j->atlasDim[1] + (size_t)q[1] ) * j->atlasDim[0] + (size_t)q[0] );
        float there[4] = { 0, 0, 0, 0 };
        orc_tf_fetch( j->tf, d2 * multiplyer + addedValue, j->opt.tfFracBits, tfn );
        orc_composite( tfn, there, alphaCorrection );
        for( int k = 0; k < 4; ++k )
            worst = fmaxf( worst, fabsf( there[k] - here[k] ) );
    }
    return worst * transmittance;
}

/* TEST INSTRUMENT (orc_options.tieBudget), second part: a brick the ray grazes.  Where a ray passes a brick
 * edge or corner, the slab test of a brick it merely touches comes out with tfar within rounding of tnear:
 * a last bit decides whether that brick gets its one sample (Renderer.cu:79, :208) -- and an implementation
 * whose ray differs in the last bit (another, equally valid evaluation of the matrices or of the ray set-up)
 * decides differently.  Returns the weight such a sample would have (classified alpha of the voxel at the
 * touching point, times the transmittance), 0 when the interval is not degenerate. */
static float sliver_budget( const job_t* j, const orc_node_data* nodeData, f3 origin, f3 dir, float tNear,
                            float tFar, float multiplyer, float addedValue, float alphaCorrection,
                            float transmittance )
{
    const float scale = fmaxf( 1.0f, fabsf( tNear ) );
    if( !( fabsf( tFar - tNear ) <= 4e-6f * scale ) )
        return 0.0f;
    const f3 pos = { origin.x + dir.x * tNear, origin.y + dir.y * tNear, origin.z + dir.z * tNear };
    const f3 texPos = {
        ( ( pos.x - nodeData->aabbMin[0] ) / nodeData->aabbSize[0] ) * nodeData->textureSize[0] + nodeData->textureMin[0],
        ( ( pos.y - nodeData->aabbMin[1] ) / nodeData->aabbSize[1] ) * nodeData->textureSize[1] + nodeData->textureMin[1],
        ( ( pos.z - nodeData->aabbMin[2] ) / nodeData->aabbSize[2] ) * nodeData->textureSize[2] + nodeData->textureMin[2] };
    const float density = j->opt.filter ? fetch_trilinear( j->atlas, j->opt.voxelBytes, j->atlasDim, texPos )
                                        : fetch_nearest( j->atlas, j->opt.voxelBytes, j->atlasDim, texPos );
    float tfn[4], here[4] = { 0, 0, 0, 0 };
    orc_tf_fetch( j->tf, density * multiplyer + addedValue, j->opt.tfFracBits, tfn );
    orc_composite( tfn, here, alphaCorrection );
    return fmaxf( fmaxf( here[0], here[1] ), fmaxf( here[2], here[3] ) ) * transmittance;
}

/* TEST INSTRUMENT (orc_options.tieBudget), third part: the last sample of a brick segment.  The march takes
 * ceil(dist / stepSize) samples (Renderer.cu:208: travel = dist; travel > 0; travel -= stepSize); where dist is
 * within rounding of a whole number of steps, a last bit of the ray (tnear, tfar of the slab test) decides
 * whether the sample at the far face is taken.  Returns the weight of the sample at texPos. */
static float sample_weight( const job_t* j, f3 texPos, float multiplyer, float addedValue, float alphaCorrection,
                            float transmittance )
{
    const float density = j->opt.filter ? fetch_trilinear( j->atlas, j->opt.voxelBytes, j->atlasDim, texPos )
                                        : fetch_nearest( j->atlas, j->opt.voxelBytes, j->atlasDim, texPos );
    float tfn[4], here[4] = { 0, 0, 0, 0 };
    orc_tf_fetch( j->tf, density * multiplyer + addedValue, j->opt.tfFracBits, tfn );
    orc_composite( tfn, here, alphaCorrection );
    return fmaxf( fmaxf( here[0], here[1] ), fmaxf( here[2], here[3] ) ) * transmittance;
}

/* one pixel: Renderer.cu:106-229 */
static uint64_t raycast_pixel( const job_t* j, uint32_t x, uint32_t y )
{
    const orc_view_data* viewData = j->view;
    const orc_render_data* renderData = j->render;
    uint64_t nSamples = 0;

    const f4 pixelEyeSpacePos =
        eye_space_from_window( (float)x, (float)y, viewData->glViewport, viewData->invProjMatrix );
    const f4 pixelWorldSpacePos = mat_mul_vec4( viewData->invViewMatrix, pixelEyeSpacePos );
    const f3 eyePos = { viewData->eyePosition[0], viewData->eyePosition[1],
                        viewData->eyePosition[2] };
    const f3 d0 = { pixelWorldSpacePos.x - eyePos.x, pixelWorldSpacePos.y - eyePos.y,
                    pixelWorldSpacePos.z - eyePos.z };
    f3 dir = normalize_f3( d0 );
    if( dir.x == 0.0f ) dir.x = EPSILON;
    if( dir.y == 0.0f ) dir.y = EPSILON;
    if( dir.z == 0.0f ) dir.z = EPSILON;

    float tNearGlobal, tFarGlobal;
    const f3 globalBoxMin = { viewData->aabbMin[0], viewData->aabbMin[1], viewData->aabbMin[2] };
    const f3 globalBoxMax = { viewData->aabbMax[0], viewData->aabbMax[1], viewData->aabbMax[2] };
    const f3 origin = eyePos;
    if( !intersect_box( origin, dir, globalBoxMin, globalBoxMax, &tNearGlobal, &tFarGlobal ) )
        return 0;

    /* Renderer.cu:132-146 */
    for( uint32_t i = 0; i < j->nPlanes; ++i )
    {
        const float* cp = j->clipPlanes + 4 * i;
        const f3 planeNormal = { cp[0], cp[1], cp[2] };
        float rn = dot3( dir, planeNormal );
        if( rn == 0.0f )
            rn = EPSILON;
        const float d = cp[3];
        const float t = -( dot3( planeNormal, eyePos ) + d ) / rn;
        if( rn > 0.0f )
            tNearGlobal = fmaxf( tNearGlobal, t );
        else
            tFarGlobal = fminf( tFarGlobal, t );
    }
    if( tNearGlobal > tFarGlobal )
        return 0;

    const size_t pixelPos = (size_t)y * j->width + x;
    float* px = j->pixelBuffer + pixelPos * 4;
    if( px[3] > EARLY_EXIT )
        return 0;
    float color[4] = { px[0], px[1], px[2], px[3] };

    const f3 e3 = { pixelEyeSpacePos.x, pixelEyeSpacePos.y, pixelEyeSpacePos.z };
    const f3 nPixelEyeSpacePos = normalize_f3( e3 );
    const float tNearPlane = -viewData->nearPlane / nPixelEyeSpacePos.z;

    const float r0 = renderData->dataSourceRange[0], r1 = renderData->dataSourceRange[1];
    const float multiplyer = 1.0f / ( r1 - r0 );
    const float addedValue = -r0 / ( r1 - r0 );
    const float alphaCorrection =
        (float)renderData->maxSamplesPerRay / (float)renderData->samplesPerRay;
    /* Renderer.cu:170: 1.0 / float(spr) is a double division rounded to float on store */
    const float stepSize = (float)( 1.0 / (double)(float)renderData->samplesPerRay );

    /* TEST INSTRUMENT (orc_options.tieBudget), fourth part: the early-exit test itself (Renderer.cu:219-226).  Two
     * evaluations whose opacities differ by ertEps (what the parity rule allows them to: E0 + 2 x the budget so far)
     * end the ray at different samples when an opacity comes out that close to the threshold: anywhere from the
     * first sample that leaves the opacity above threshold - ertEps (ertLo = the opacity after it) to the first
     * that leaves it above threshold + ertEps.  The results differ by at most the opacity gained in between, which
     * is added to the budget; when the reference's own exit lies inside that window the march goes on on a copy
     * (shadow, acc) until the window is left.  color and nSamples stay the restated algorithm's. */
    float shadow[4] = { 0, 0, 0, 0 }, *acc = color, ertLo = -1.0f, ertEps = 0.0f;
    int shadowMode = 0;

    for( uint32_t i = 0; i < j->nodeCount; ++i )
    {
        const orc_node_data* nodeData = &j->nodes[i];
        const f3 boxMin = { nodeData->aabbMin[0], nodeData->aabbMin[1], nodeData->aabbMin[2] };
        const f3 boxSize = { nodeData->aabbSize[0], nodeData->aabbSize[1], nodeData->aabbSize[2] };
        const f3 boxMax = { boxMin.x + boxSize.x, boxMin.y + boxSize.y, boxMin.z + boxSize.z };

        float tNear = 0.0f, tFar = 0.0f;
        const int hitBox = intersect_box( origin, dir, boxMin, boxMax, &tNear, &tFar );
        if( j->opt.tieBudget && !shadowMode && tFar >= tNearGlobal && tNear <= tFarGlobal &&
            tFar >= tNearPlane ) /* test instrument, see orc_options */
            j->opt.tieBudget[pixelPos] += sliver_budget( j, nodeData, origin, dir, tNear, tFar, multiplyer,
                                                         addedValue, alphaCorrection, 1.0f - color[3] );
        if( !hitBox )
            continue;
        if( tNear > tFarGlobal )
            break;
        if( tFar < tNearGlobal )
            continue;
        tNear = fmaxf( fmaxf( tNearPlane, tNear ), tNearGlobal );
        tFar = fminf( tFar, tFarGlobal );
        if( tNear > tFar )
            continue;

        const f3 rayStart = { origin.x + dir.x * tNear, origin.y + dir.y * tNear,
                              origin.z + dir.z * tNear };
        const f3 rayStop = { origin.x + dir.x * tFar, origin.y + dir.y * tFar,
                             origin.z + dir.z * tFar };
        f3 pos = rayStart;
        if( j->opt.entryBias != 0.0f ) /* test instrument (orc_options.entryBias), not the reference */
        {
            pos.x -= dir.x * j->opt.entryBias;
            pos.y -= dir.y * j->opt.entryBias;
            pos.z -= dir.z * j->opt.entryBias;
        }
        const f3 diff = { rayStop.x - rayStart.x, rayStop.y - rayStart.y, rayStop.z - rayStart.z };
        const f3 ndiff = normalize_f3( diff );
        const f3 step = { ndiff.x * stepSize, ndiff.y * stepSize, ndiff.z * stepSize };
        const float dist = sqrtf( dot3( diff, diff ) ); /* math.cuh:1522-1526 */

        const f3 texMin = { nodeData->textureMin[0], nodeData->textureMin[1], nodeData->textureMin[2] };
        const f3 texSize = { nodeData->textureSize[0], nodeData->textureSize[1], nodeData->textureSize[2] };

        int isEarlyExit = 0;
        uint32_t kStep = 0; /* test instrument only (tie budget) */
        const f3 vpw = { texSize.x * (float)j->atlasDim[0] / boxSize.x, texSize.y * (float)j->atlasDim[1] / boxSize.y,
                         texSize.z * (float)j->atlasDim[2] / boxSize.z };
        /* test instrument only: how far a last bit of the ray moves the end of the segment (as sliver_budget) */
        const float endEps = 4e-6f * fmaxf( 1.0f, fabsf( tFar ) );
        float travel;
        for( travel = dist; travel > 0.0f;
             pos.x += step.x, pos.y += step.y, pos.z += step.z, travel -= stepSize, ++kStep )
        {
            const f3 texPos = { ( ( pos.x - boxMin.x ) / boxSize.x ) * texSize.x + texMin.x,
                                ( ( pos.y - boxMin.y ) / boxSize.y ) * texSize.y + texMin.y,
                                ( ( pos.z - boxMin.z ) / boxSize.z ) * texSize.z + texMin.z };
            const float density = j->opt.filter
                                      ? fetch_trilinear( j->atlas, j->opt.voxelBytes, j->atlasDim, texPos )
                                      : fetch_nearest( j->atlas, j->opt.voxelBytes, j->atlasDim, texPos );
            float transferFn[4];
            if( j->opt.tieBudget && !shadowMode ) /* test instrument, see orc_options */
            {
                if( !j->opt.filter ) /* a voxel flip is a point-sampling matter; whether a sample exists is not */
                    j->opt.tieBudget[pixelPos] += tie_budget( j, texPos, density, multiplyer, addedValue,
                                                              alphaCorrection, 1.0f - color[3], kStep, vpw );
                if( travel <= endEps && kStep > 0 ) /* the sample at the far face barely made it */
                    j->opt.tieBudget[pixelPos] += sample_weight( j, texPos, multiplyer, addedValue, alphaCorrection,
                                                                 1.0f - color[3] );
            }
            orc_tf_fetch( j->tf, density * multiplyer + addedValue, j->opt.tfFracBits, transferFn );
            orc_composite( transferFn, acc, alphaCorrection );
            if( !shadowMode )
            {
                ++nSamples;
                isEarlyExit = color[3] > EARLY_EXIT;
            }
            if( j->opt.tieBudget ) /* test instrument, fourth part: see ertLo below */
            {
                if( ertLo < 0.0f )
                {
                    ertEps = 5e-5f + 2.0f * j->opt.tieBudget[pixelPos];
                    if( acc[3] > EARLY_EXIT - ertEps )
                        ertLo = acc[3];
                }
                if( shadowMode )
                {
                    if( acc[3] > EARLY_EXIT + ertEps )
                        break;
                    continue;
                }
                if( isEarlyExit && color[3] <= EARLY_EXIT + ertEps )
                {
                    /* the reference's ray ends here; an evaluation whose opacity is ertEps lower goes on: follow
                     * it on a copy until it, too, must have ended */
                    shadowMode = 1;
                    shadow[0] = color[0], shadow[1] = color[1], shadow[2] = color[2], shadow[3] = color[3];
                    acc = shadow;
                    continue;
                }
            }
            if( isEarlyExit )
                break;
        }
        if( shadowMode && shadow[3] > EARLY_EXIT + ertEps )
            break;
        if( isEarlyExit && !shadowMode )
            break;
        if( j->opt.tieBudget && !shadowMode && travel > -endEps && kStep > 0 ) /* ... or barely did not */
        {
            const f3 texPos = { ( ( pos.x - boxMin.x ) / boxSize.x ) * texSize.x + texMin.x,
                                ( ( pos.y - boxMin.y ) / boxSize.y ) * texSize.y + texMin.y,
                                ( ( pos.z - boxMin.z ) / boxSize.z ) * texSize.z + texMin.z };
            j->opt.tieBudget[pixelPos] += sample_weight( j, texPos, multiplyer, addedValue, alphaCorrection,
                                                         1.0f - color[3] );
        }
    }
    if( j->opt.tieBudget && ertLo >= 0.0f )
        j->opt.tieBudget[pixelPos] += fmaxf( 0.0f, acc[3] - ertLo );
    px[0] = color[0];
    px[1] = color[1];
    px[2] = color[2];
    px[3] = color[3];
    return nSamples;
}


/* rand() of fragRaycast.glsl:59-62.  sin() of arguments up to ~1e6 is implementation-defined in GLSL to more
 * bits than the factor 43758.5453 leaves, so the jitter is "unpinned" by construction; this restatement fixes it:
 * dot product and the rest in float, the sine in double (vrc_gl_rand in vrc_core.h is the same definition). */
static float gl_rand( float x, float y )
{
    const float d = x * 12.9898f + y * 78.233f;
    const float sn = (float)sin( (double)d );
    const float v = sn * 43758.5453f;
    return v - floorf( v );
}

/* glRaycaster with nSamplesPerPixel > 1 (fragRaycast.glsl:113-215): one fragment per pixel and BRICK; for each of
 * the n jittered sub-pixel rays the brick is marched starting from the pixel's colour so far (localResult = result,
 * :125), the pixel becomes the average of the n results (:212-214).  Any `discard` inside the sub-sample loop
 * (:140-143, :159-160, :176-177) discards the whole fragment: that brick leaves the pixel unchanged.  The tie budget
 * (test instrument) gets each sub-ray's ties divided by n; the parts that follow a single ray past its early exit
 * are not carried over (the tests of this mode use transfer functions that do not reach the threshold). */
static uint64_t raycast_pixel_gl_ss( const job_t* j, uint32_t x, uint32_t y )
{
    const orc_view_data* viewData = j->view;
    const orc_render_data* renderData = j->render;
    uint64_t nSamples = 0;
    const uint32_t n = renderData->samplesPerPixel;
    const size_t pixelPos = (size_t)y * j->width + x;
    float* px = j->pixelBuffer + pixelPos * 4;
    float color[4] = { px[0], px[1], px[2], px[3] };
    const float fx = (float)x + 0.5f, fy = (float)y + 0.5f; /* gl_FragCoord.xy */
    const f3 eyePos = { viewData->eyePosition[0], viewData->eyePosition[1], viewData->eyePosition[2] };
    const f3 origin = eyePos;
    const f3 globalBoxMin = { viewData->aabbMin[0], viewData->aabbMin[1], viewData->aabbMin[2] };
    const f3 globalBoxMax = { viewData->aabbMax[0], viewData->aabbMax[1], viewData->aabbMax[2] };
    const float r0 = renderData->dataSourceRange[0], r1 = renderData->dataSourceRange[1];
    const float multiplyer = 1.0f / ( r1 - r0 );
    const float addedValue = -r0 / ( r1 - r0 );
    const float alphaCorrection = (float)renderData->maxSamplesPerRay / (float)renderData->samplesPerRay;
    const float stepSize = 1.0f / (float)renderData->samplesPerRay;

    for( uint32_t i = 0; i < j->nodeCount; ++i )
    {
        if( color[3] > EARLY_EXIT ) /* :115-117 */
            break;
        const orc_node_data* nodeData = &j->nodes[i];
        const f3 boxMin = { nodeData->aabbMin[0], nodeData->aabbMin[1], nodeData->aabbMin[2] };
        const f3 boxSize = { nodeData->aabbSize[0], nodeData->aabbSize[1], nodeData->aabbSize[2] };
        const f3 boxMax = { boxMin.x + boxSize.x, boxMin.y + boxSize.y, boxMin.z + boxSize.z };
        const f3 texMin = { nodeData->textureMin[0], nodeData->textureMin[1], nodeData->textureMin[2] };
        const f3 texSize = { nodeData->textureSize[0], nodeData->textureSize[1], nodeData->textureSize[2] };
        const f3 vpw = { texSize.x * (float)j->atlasDim[0] / boxSize.x, texSize.y * (float)j->atlasDim[1] / boxSize.y,
                         texSize.z * (float)j->atlasDim[2] / boxSize.z };
        float brickResult[4] = { 0.f, 0.f, 0.f, 0.f };
        float budget = 0.0f;
        uint64_t cnt = 0;
        int discard = 0;
        for( uint32_t k = 0; k < n && !discard; ++k )
        {
            const float fk = (float)k;
            const float xPixelDelta = gl_rand( fx * fk, fy * fk ) / 2.0f;                     /* :123 */
            const float yPixelDelta = gl_rand( fx * 2.0f * fk, fy * 2.0f * fk ) / 2.0f;       /* :124 */
            float localResult[4] = { color[0], color[1], color[2], color[3] };               /* :125 */
            const f4 pixelEyeSpacePos = eye_space_from_window( fx + xPixelDelta, fy + yPixelDelta,
                                                               viewData->glViewport, viewData->invProjMatrix );
            const f4 pixelWorldSpacePos = mat_mul_vec4( viewData->invViewMatrix, pixelEyeSpacePos );
            const f3 d0 = { pixelWorldSpacePos.x - eyePos.x, pixelWorldSpacePos.y - eyePos.y,
                            pixelWorldSpacePos.z - eyePos.z };
            f3 dir = normalize_f3( d0 );
            if( dir.x == 0.0f ) dir.x = EPSILON;
            if( dir.y == 0.0f ) dir.y = EPSILON;
            if( dir.z == 0.0f ) dir.z = EPSILON;
            float tnearGlobal, tfarGlobal;
            intersect_box( origin, dir, globalBoxMin, globalBoxMax, &tnearGlobal, &tfarGlobal );
            if( !( tnearGlobal <= tfarGlobal ) ) { discard = 1; break; }                      /* :139-140 */
            float tnear = 0.0f, tfar = 0.0f;
            intersect_box( origin, dir, boxMin, boxMax, &tnear, &tfar );
            if( !( tnear <= tfar ) ) { discard = 1; break; }                                  /* :142-143 */
            const f3 e3 = { pixelEyeSpacePos.x, pixelEyeSpacePos.y, pixelEyeSpacePos.z };
            const f3 nPixelEyeSpacePos = normalize_f3( e3 );
            const float tNearPlane = -viewData->nearPlane / nPixelEyeSpacePos.z;
            if( tnear < tNearPlane )
                tnear = tNearPlane;
            const float a = tnear - tnearGlobal;
            const float residu = a - stepSize * floorf( a / stepSize ); /* GLSL mod() */
            if( residu > 0.0f )
                tnear += stepSize - residu;
            if( tnear > tfar ) { discard = 1; break; }                                        /* :159-160 */
            for( uint32_t p = 0; p < j->nPlanes; ++p )
            {
                const float* cp = j->clipPlanes + 4 * p;
                const f3 planeNormal = { cp[0], cp[1], cp[2] };
                float rn = dot3( dir, planeNormal );
                if( rn == 0.0f )
                    rn = EPSILON;
                const float t = -( dot3( planeNormal, eyePos ) + cp[3] ) / rn;
                if( rn > 0.0f )
                    tnear = fmaxf( tnear, t );
                else
                    tfar = fminf( tfar, t );
            }
            if( tnear > tfar ) { discard = 1; break; }                                        /* :176-177 */
            const f3 rayStart = { origin.x + dir.x * tnear, origin.y + dir.y * tnear, origin.z + dir.z * tnear };
            const f3 rayStop = { origin.x + dir.x * tfar, origin.y + dir.y * tfar, origin.z + dir.z * tfar };
            f3 pos = rayStart;
            const f3 diff = { rayStop.x - rayStart.x, rayStop.y - rayStart.y, rayStop.z - rayStart.z };
            const f3 ndiff = normalize_f3( diff );
            const f3 step = { ndiff.x * stepSize, ndiff.y * stepSize, ndiff.z * stepSize };
            const float dist = sqrtf( dot3( diff, diff ) );
            const float endEps = 4e-6f * fmaxf( 1.0f, fabsf( tfar ) );
            uint32_t kStep = 0;
            float travel;
            for( travel = dist; travel > 0.0f;
                 pos.x += step.x, pos.y += step.y, pos.z += step.z, travel -= stepSize, ++kStep )
            {
                const f3 texPos = { ( ( pos.x - boxMin.x ) / boxSize.x ) * texSize.x + texMin.x,
                                    ( ( pos.y - boxMin.y ) / boxSize.y ) * texSize.y + texMin.y,
                                    ( ( pos.z - boxMin.z ) / boxSize.z ) * texSize.z + texMin.z };
                const float density = j->opt.filter
                                          ? fetch_trilinear( j->atlas, j->opt.voxelBytes, j->atlasDim, texPos )
                                          : fetch_nearest( j->atlas, j->opt.voxelBytes, j->atlasDim, texPos );
                if( j->opt.tieBudget ) /* test instrument, see orc_options */
                {
                    if( !j->opt.filter )
                        budget += tie_budget( j, texPos, density, multiplyer, addedValue, alphaCorrection,
                                              1.0f - localResult[3], kStep, vpw );
                    if( travel <= endEps && kStep > 0 )
                        budget += sample_weight( j, texPos, multiplyer, addedValue, alphaCorrection,
                                                 1.0f - localResult[3] );
                }
                float transferFn[4];
                orc_tf_fetch( j->tf, density * multiplyer + addedValue, j->opt.tfFracBits, transferFn );
                orc_composite( transferFn, localResult, alphaCorrection );
                ++cnt;
                if( localResult[3] > EARLY_EXIT ) /* :205-206 */
                    break;
            }
            if( j->opt.tieBudget && travel > -endEps && travel <= 0.0f && kStep > 0 ) /* a sample that barely was not taken */
            {
                const f3 texPos = { ( ( pos.x - boxMin.x ) / boxSize.x ) * texSize.x + texMin.x,
                                    ( ( pos.y - boxMin.y ) / boxSize.y ) * texSize.y + texMin.y,
                                    ( ( pos.z - boxMin.z ) / boxSize.z ) * texSize.z + texMin.z };
                budget += sample_weight( j, texPos, multiplyer, addedValue, alphaCorrection, 1.0f - localResult[3] );
            }
            brickResult[0] += localResult[0];
            brickResult[1] += localResult[1];
            brickResult[2] += localResult[2];
            brickResult[3] += localResult[3];
        }
        if( discard )
            continue;
        const float fn = (float)n;
        color[0] = brickResult[0] / fn; /* :214 */
        color[1] = brickResult[1] / fn;
        color[2] = brickResult[2] / fn;
        color[3] = brickResult[3] / fn;
        nSamples += cnt;
        if( j->opt.tieBudget )
            j->opt.tieBudget[pixelPos] += budget / fn;
    }
    px[0] = color[0];
    px[1] = color[1];
    px[2] = color[2];
    px[3] = color[3];
    return nSamples;
}

/* one pixel of the GLSL twin, renderers/glRaycaster/shaders/fragRaycast.glsl:113-215, with the
 * per-brick draws (GLRaycastRenderer.cpp:431-510) folded into one loop over the sorted bricks:
 * the accumulation image (imageLoad/imageStore, :115, :214) is the running colour; a brick the
 * ray misses or that lies behind the early-exit threshold is "discard".  nSamplesPerPixel = 1
 * (jitter rand(0,0)/2 = 0, :121-127; more samples per pixel: raycast_pixel_gl_ss above).  Differences from
 * Renderer.cu, line by line:
 *   :127     gl_FragCoord = pixel centre (x+0.5, y+0.5), the CUDA kernel uses (x, y)
 *   :101     intersectBox returns t0 <= t1 (CUDA: tfar > tnear)
 *   :149-150 tnear is raised to the near plane only (CUDA also clamps to the global interval)
 *   :154-157 residu = mod(tnear - tnearGlobal, stepSize); tnear += stepSize - residu if > 0
 *   :162-174 clip planes move this brick's tnear/tfar, after the snap
 * The volume fetch is the atlas fetch of the CUDA variant (the GL renderer keeps one GL_NEAREST
 * texture per brick, TexturePool.cpp:103-104: same voxel up to coordinate rounding); the transfer
 * function an RGBA8 texture (GLRaycastRenderer.cpp:188-192): the caller passes it quantised. */
static uint64_t raycast_pixel_gl( const job_t* j, uint32_t x, uint32_t y )
{
    const orc_view_data* viewData = j->view;
    const orc_render_data* renderData = j->render;
    uint64_t nSamples = 0;

    const size_t pixelPos = (size_t)y * j->width + x;
    float* px = j->pixelBuffer + pixelPos * 4;
    float color[4] = { px[0], px[1], px[2], px[3] };

    const f4 pixelEyeSpacePos = eye_space_from_window( (float)x + 0.5f, (float)y + 0.5f,
                                                       viewData->glViewport, viewData->invProjMatrix );
    const f4 pixelWorldSpacePos = mat_mul_vec4( viewData->invViewMatrix, pixelEyeSpacePos );
    const f3 eyePos = { viewData->eyePosition[0], viewData->eyePosition[1],
                        viewData->eyePosition[2] };
    const f3 d0 = { pixelWorldSpacePos.x - eyePos.x, pixelWorldSpacePos.y - eyePos.y,
                    pixelWorldSpacePos.z - eyePos.z };
    f3 dir = normalize_f3( d0 );
    if( dir.x == 0.0f ) dir.x = EPSILON;
    if( dir.y == 0.0f ) dir.y = EPSILON;
    if( dir.z == 0.0f ) dir.z = EPSILON;
    const f3 origin = eyePos;

    float tnearGlobal, tfarGlobal;
    const f3 globalBoxMin = { viewData->aabbMin[0], viewData->aabbMin[1], viewData->aabbMin[2] };
    const f3 globalBoxMax = { viewData->aabbMax[0], viewData->aabbMax[1], viewData->aabbMax[2] };
    intersect_box( origin, dir, globalBoxMin, globalBoxMax, &tnearGlobal, &tfarGlobal );
    if( !( tnearGlobal <= tfarGlobal ) )
        return 0;

    const f3 e3 = { pixelEyeSpacePos.x, pixelEyeSpacePos.y, pixelEyeSpacePos.z };
    const f3 nPixelEyeSpacePos = normalize_f3( e3 );
    const float tNearPlane = -viewData->nearPlane / nPixelEyeSpacePos.z;

    const float r0 = renderData->dataSourceRange[0], r1 = renderData->dataSourceRange[1];
    const float multiplyer = 1.0f / ( r1 - r0 );
    const float addedValue = -r0 / ( r1 - r0 );
    const float alphaCorrection =
        (float)renderData->maxSamplesPerRay / (float)renderData->samplesPerRay;
    const float stepSize = 1.0f / (float)renderData->samplesPerRay;

    /* test instrument, parts three and four as in raycast_pixel: the last sample of a segment and the early-exit test */
    float shadow[4] = { 0, 0, 0, 0 }, *acc = color, ertLo = -1.0f, ertEps = 0.0f;
    int shadowMode = 0;

    for( uint32_t i = 0; i < j->nodeCount; ++i )
    {
        if( !shadowMode && color[3] > EARLY_EXIT ) /* :115-117 */
        {
            if( j->opt.tieBudget && color[3] <= EARLY_EXIT + 5e-5f + 2.0f * j->opt.tieBudget[pixelPos] )
            {
                if( ertLo < 0.0f )
                {
                    ertEps = 5e-5f + 2.0f * j->opt.tieBudget[pixelPos];
                    ertLo = color[3];
                }
                shadowMode = 1;
                shadow[0] = color[0], shadow[1] = color[1], shadow[2] = color[2], shadow[3] = color[3];
                acc = shadow;
            }
            else
                break;
        }
        if( shadowMode && shadow[3] > EARLY_EXIT + ertEps )
            break;
        const orc_node_data* nodeData = &j->nodes[i];
        const f3 boxMin = { nodeData->aabbMin[0], nodeData->aabbMin[1], nodeData->aabbMin[2] };
        const f3 boxSize = { nodeData->aabbSize[0], nodeData->aabbSize[1], nodeData->aabbSize[2] };
        const f3 boxMax = { boxMin.x + boxSize.x, boxMin.y + boxSize.y, boxMin.z + boxSize.z };

        float tnear = 0.0f, tfar = 0.0f;
        intersect_box( origin, dir, boxMin, boxMax, &tnear, &tfar );
        if( !( tnear <= tfar ) )
            continue;
        if( tnear < tNearPlane )
            tnear = tNearPlane;
        const float a = tnear - tnearGlobal;
        const float residu = a - stepSize * floorf( a / stepSize ); /* GLSL mod() */
        if( residu > 0.0f )
            tnear += stepSize - residu;
        if( tnear > tfar )
            continue;
        for( uint32_t k = 0; k < j->nPlanes; ++k )
        {
            const float* cp = j->clipPlanes + 4 * k;
            const f3 planeNormal = { cp[0], cp[1], cp[2] };
            float rn = dot3( dir, planeNormal );
            if( rn == 0.0f )
                rn = EPSILON;
            const float t = -( dot3( planeNormal, eyePos ) + cp[3] ) / rn;
            if( rn > 0.0f )
                tnear = fmaxf( tnear, t );
            else
                tfar = fminf( tfar, t );
        }
        if( tnear > tfar )
            continue;

        const f3 rayStart = { origin.x + dir.x * tnear, origin.y + dir.y * tnear,
                              origin.z + dir.z * tnear };
        const f3 rayStop = { origin.x + dir.x * tfar, origin.y + dir.y * tfar,
                             origin.z + dir.z * tfar };
        f3 pos = rayStart;
        const f3 diff = { rayStop.x - rayStart.x, rayStop.y - rayStart.y, rayStop.z - rayStart.z };
        const f3 ndiff = normalize_f3( diff );
        const f3 step = { ndiff.x * stepSize, ndiff.y * stepSize, ndiff.z * stepSize };
        const float dist = sqrtf( dot3( diff, diff ) );
        const f3 texMin = { nodeData->textureMin[0], nodeData->textureMin[1], nodeData->textureMin[2] };
        const f3 texSize = { nodeData->textureSize[0], nodeData->textureSize[1], nodeData->textureSize[2] };

        uint32_t kStep = 0; /* test instrument only (tie budget) */
        const f3 vpw = { texSize.x * (float)j->atlasDim[0] / boxSize.x, texSize.y * (float)j->atlasDim[1] / boxSize.y,
                         texSize.z * (float)j->atlasDim[2] / boxSize.z };
        const float endEps = 4e-6f * fmaxf( 1.0f, fabsf( tfar ) );
        int stopHere = 0;
        float travel;
        for( travel = dist; travel > 0.0f;
             pos.x += step.x, pos.y += step.y, pos.z += step.z, travel -= stepSize, ++kStep )
        {
            const f3 texPos = { ( ( pos.x - boxMin.x ) / boxSize.x ) * texSize.x + texMin.x,
                                ( ( pos.y - boxMin.y ) / boxSize.y ) * texSize.y + texMin.y,
                                ( ( pos.z - boxMin.z ) / boxSize.z ) * texSize.z + texMin.z };
            const float density = j->opt.filter
                                      ? fetch_trilinear( j->atlas, j->opt.voxelBytes, j->atlasDim, texPos )
                                      : fetch_nearest( j->atlas, j->opt.voxelBytes, j->atlasDim, texPos );
            float transferFn[4];
            if( j->opt.tieBudget && !shadowMode ) /* test instrument, see orc_options */
            {
                if( !j->opt.filter )
                    j->opt.tieBudget[pixelPos] += tie_budget( j, texPos, density, multiplyer, addedValue,
                                                              alphaCorrection, 1.0f - color[3], kStep, vpw );
                if( travel <= endEps && kStep > 0 )
                    j->opt.tieBudget[pixelPos] += sample_weight( j, texPos, multiplyer, addedValue, alphaCorrection,
                                                                 1.0f - color[3] );
            }
            orc_tf_fetch( j->tf, density * multiplyer + addedValue, j->opt.tfFracBits, transferFn );
            orc_composite( transferFn, acc, alphaCorrection );
            if( !shadowMode )
                ++nSamples;
            if( j->opt.tieBudget )
            {
                if( ertLo < 0.0f )
                {
                    ertEps = 5e-5f + 2.0f * j->opt.tieBudget[pixelPos];
                    if( acc[3] > EARLY_EXIT - ertEps )
                        ertLo = acc[3];
                }
                if( shadowMode )
                {
                    if( acc[3] > EARLY_EXIT + ertEps )
                    {
                        stopHere = 1;
                        break;
                    }
                    continue;
                }
                if( color[3] > EARLY_EXIT && color[3] <= EARLY_EXIT + ertEps )
                {
                    shadowMode = 1;
                    shadow[0] = color[0], shadow[1] = color[1], shadow[2] = color[2], shadow[3] = color[3];
                    acc = shadow;
                    continue;
                }
            }
            if( color[3] > EARLY_EXIT )
            {
                stopHere = 1;
                break;
            }
        }
        if( stopHere )
            continue; /* the test at the top of the brick loop ends the ray */
        if( j->opt.tieBudget && !shadowMode && travel > -endEps && kStep > 0 )
        {
            const f3 texPos = { ( ( pos.x - boxMin.x ) / boxSize.x ) * texSize.x + texMin.x,
                                ( ( pos.y - boxMin.y ) / boxSize.y ) * texSize.y + texMin.y,
                                ( ( pos.z - boxMin.z ) / boxSize.z ) * texSize.z + texMin.z };
            j->opt.tieBudget[pixelPos] += sample_weight( j, texPos, multiplyer, addedValue, alphaCorrection,
                                                         1.0f - color[3] );
        }
    }
    if( j->opt.tieBudget && ertLo >= 0.0f )
        j->opt.tieBudget[pixelPos] += fmaxf( 0.0f, acc[3] - ertLo );
    px[0] = color[0];
    px[1] = color[1];
    px[2] = color[2];
    px[3] = color[3];
    return nSamples;
}

/* ------------------------------------------------------------------------------------------
 * EXTENSION: per-ray adaptive LOD (BASELINE C5).  Not in the reference, which selects the LOD per
 * brick on the host: SelectVisibles.cpp:52-68 calls a brick fine enough when
 *     worldSpacePerVoxel / worldSpacePerPixel * near / (near + distance) <= screenSpaceError
 * at the point of its box nearest to the near plane.  Definition used here (the HIP kernel's
 * vrc_pixel_ray_lod must reproduce it):
 *   - nodes form a hierarchy.  Voxel size of a node = aabbSize.x / (textureSize.x * atlasDim.x);
 *     sizes within 5 % of each other are one level, levels numbered from the finest (0); a level
 *     is a regular grid of bricks of its largest box size anchored at the min corner of all
 *     boxes (border bricks may be smaller: ragged trees such as UVF's, whose levels do not
 *     align with each other, are fine), one brick per grid cell at most;
 *   - along a ray eye-space depth / near = t / tNearPlane, so level j is fine enough from
 *     T_j = tNearPlane * lodBase * 2^j, lodBase = finest voxel size / (sse * worldSpacePerPixel);
 *   - the ray hops from brick to brick.  At parameter te it wants level
 *     k = #{ j in 1..K-1 : T_j <= te } and takes, at the point o + d*(te + eps), the brick of the
 *     first level that has one there in the order k, k+1, .., K-1, k-1, .., 0; the run ends where
 *     the ray leaves that brick's box (or the finest level's grid cell where there is no brick),
 *     and the level is chosen anew there.  eps = 1 % of the finest voxel;
 *   - a run is integrated like a reference brick segment (Renderer.cu:195-223) over
 *     [te + eps, exit] (the first sample lies eps inside the brick, not on its face, where the
 *     voxel it reads would hang on the last bit of te) with stepSize * 2^level and opacity
 *     exponent alphaCorrection * 2^level.
 * ---------------------------------------------------------------------------------------- */
#define LOD_MAX_LEVELS 8
typedef struct lod_grid
{
    int K;
    int dim[LOD_MAX_LEVELS][3];
    float gmin[3], gmax[3];
    float cell[LOD_MAX_LEVELS][3], invCell[LOD_MAX_LEVELS][3];
    size_t offset[LOD_MAX_LEVELS];
    float lodBase, eps;
    int32_t* tables; /* per level dim.x*dim.y*dim.z entries at offset[level], -1 = none */
    uint8_t* level;  /* per node */
} lod_grid;

static int near_int( double v, double tol, long* out )
{
    const double r = floor( v + 0.5 );
    *out = (long)r;
    return fabs( v - r ) <= tol;
}

static void lod_grid_free( lod_grid* g )
{
    if( !g )
        return;
    free( g->tables );
    free( g->level );
    free( g );
}

static lod_grid* lod_grid_build( const orc_node_data* nodes, uint32_t n, const uint32_t atlasDim[3],
                                 float sse, float worldPerPixel )
{
    if( n == 0 )
        return NULL;
    lod_grid* g = (lod_grid*)calloc( 1, sizeof( lod_grid ) );
    double* vw = (double*)malloc( n * sizeof( double ) );
    g->level = (uint8_t*)malloc( n );
    double gmin[3], gmax[3], vw0 = 0.0;
    for( uint32_t i = 0; i < n; ++i )
    {
        const double texVox = floor( (double)nodes[i].textureSize[0] * atlasDim[0] + 0.5 );
        vw[i] = (double)nodes[i].aabbSize[0] / texVox;
        vw0 = i == 0 ? vw[i] : fmin( vw0, vw[i] );
        for( int a = 0; a < 3; ++a )
        {
            const double lo = nodes[i].aabbMin[a], hi = lo + (double)nodes[i].aabbSize[a];
            gmin[a] = i == 0 ? lo : fmin( gmin[a], lo );
            gmax[a] = i == 0 ? hi : fmax( gmax[a], hi );
        }
    }
    /* levels: clusters of voxel sizes, finest first */
    double rep[LOD_MAX_LEVELS];
    int ok = 1;
    g->K = 0;
    for( ;; )
    {
        /* the smallest size not within 5 % of (or below) the last level's */
        double next = 0.0;
        for( uint32_t i = 0; i < n; ++i )
            if( ( g->K == 0 || vw[i] > rep[g->K - 1] * 1.05 ) && ( next == 0.0 || vw[i] < next ) )
                next = vw[i];
        if( next == 0.0 )
            break;
        if( g->K == LOD_MAX_LEVELS )
        {
            ok = 0;
            break;
        }
        rep[g->K++] = next;
    }
    double cell[LOD_MAX_LEVELS][3];
    memset( cell, 0, sizeof( cell ) );
    for( uint32_t i = 0; i < n && ok; ++i )
    {
        int lv = 0;
        while( lv + 1 < g->K && vw[i] > rep[lv] * 1.05 )
            ++lv;
        g->level[i] = (uint8_t)lv;
        for( int a = 0; a < 3; ++a )
            cell[lv][a] = fmax( cell[lv][a], (double)nodes[i].aabbSize[a] );
    }
    size_t total = 0;
    for( int lv = 0; lv < g->K && ok; ++lv )
        for( int a = 0; a < 3 && ok; ++a )
        {
            ok = cell[lv][a] > 0.0;
            if( !ok )
                break;
            const double cnt = ceil( ( gmax[a] - gmin[a] ) / cell[lv][a] - 1e-3 );
            ok = cnt >= 1.0 && cnt <= 4096.0;
            g->dim[lv][a] = (int)cnt;
            g->cell[lv][a] = (float)cell[lv][a];
            g->invCell[lv][a] = (float)( 1.0 / cell[lv][a] );
            if( a == 2 )
            {
                g->offset[lv] = total;
                total += (size_t)g->dim[lv][0] * g->dim[lv][1] * g->dim[lv][2];
            }
        }
    if( ok )
    {
        g->tables = (int32_t*)malloc( total * sizeof( int32_t ) );
        for( size_t i = 0; i < total; ++i )
            g->tables[i] = -1;
    }
    for( uint32_t i = 0; i < n && ok; ++i )
    {
        const int lv = g->level[i];
        long idx[3];
        for( int a = 0; a < 3 && ok; ++a )
            ok = near_int( ( (double)nodes[i].aabbMin[a] - gmin[a] ) / cell[lv][a], 1e-3, &idx[a] ) &&
                 idx[a] >= 0 && idx[a] < g->dim[lv][a];
        if( !ok )
            break;
        int32_t* c = &g->tables[g->offset[lv] +
                                ( (size_t)idx[2] * g->dim[lv][1] + idx[1] ) * g->dim[lv][0] + idx[0]];
        if( *c != -1 ) /* two bricks of one level in one cell */
            ok = 0;
        *c = (int32_t)i;
    }
    free( vw );
    if( !ok )
    {
        lod_grid_free( g );
        return NULL;
    }
    for( int a = 0; a < 3; ++a )
    {
        g->gmin[a] = (float)gmin[a];
        g->gmax[a] = (float)gmax[a];
    }
    g->lodBase = (float)( vw0 / ( (double)sse * (double)worldPerPixel ) );
    g->eps = (float)( vw0 * 0.01 );
    return g;
}

/* test instrument (orc_options.tieBudget), part four for the per-ray LOD march: see raycast_pixel */
typedef struct
{
    float shadow[4], lo, eps;
    int on;
} ert_probe;

/* integrate one run [tA, tB] of the ray through one brick; returns 1 when the march is over (early exit) */
static int integrate_run( const job_t* j, const orc_node_data* nodeData, int level, f3 origin, f3 dir,
                          float tA, float tB, float color[4], uint64_t* nSamples, size_t pixelPos, ert_probe* ep )
{
    const orc_render_data* renderData = j->render;
    const float r0 = renderData->dataSourceRange[0], r1 = renderData->dataSourceRange[1];
    const float multiplyer = 1.0f / ( r1 - r0 );
    const float addedValue = -r0 / ( r1 - r0 );
    const float scale = (float)( 1u << level );
    const float alphaCorrection =
        (float)renderData->maxSamplesPerRay / (float)renderData->samplesPerRay * scale;
    const float stepSize = (float)( 1.0 / (double)(float)renderData->samplesPerRay ) * scale;

    const f3 boxMin = { nodeData->aabbMin[0], nodeData->aabbMin[1], nodeData->aabbMin[2] };
    const f3 boxSize = { nodeData->aabbSize[0], nodeData->aabbSize[1], nodeData->aabbSize[2] };
    const f3 texMin = { nodeData->textureMin[0], nodeData->textureMin[1], nodeData->textureMin[2] };
    const f3 texSize = { nodeData->textureSize[0], nodeData->textureSize[1], nodeData->textureSize[2] };
    const f3 rayStart = { origin.x + dir.x * tA, origin.y + dir.y * tA, origin.z + dir.z * tA };
    const f3 rayStop = { origin.x + dir.x * tB, origin.y + dir.y * tB, origin.z + dir.z * tB };
    const f3 diff = { rayStop.x - rayStart.x, rayStop.y - rayStart.y, rayStop.z - rayStart.z };
    const float d2 = dot3( diff, diff );
    if( !( d2 > 0.0f ) )
        return 0;
    const float dist = sqrtf( d2 );
    const float inv = 1.0f / dist;
    const f3 step = { diff.x * inv * stepSize, diff.y * inv * stepSize, diff.z * inv * stepSize };
    f3 pos = rayStart;
    uint32_t kStep = 0; /* test instrument only (tie budget) */
    const f3 vpw = { texSize.x * (float)j->atlasDim[0] / boxSize.x, texSize.y * (float)j->atlasDim[1] / boxSize.y,
                     texSize.z * (float)j->atlasDim[2] / boxSize.z };
    for( float travel = dist; travel > 0.0f;
         pos.x += step.x, pos.y += step.y, pos.z += step.z, travel -= stepSize, ++kStep )
    {
        const f3 texPos = { ( ( pos.x - boxMin.x ) / boxSize.x ) * texSize.x + texMin.x,
                            ( ( pos.y - boxMin.y ) / boxSize.y ) * texSize.y + texMin.y,
                            ( ( pos.z - boxMin.z ) / boxSize.z ) * texSize.z + texMin.z };
        const float density = j->opt.filter
                                  ? fetch_trilinear( j->atlas, j->opt.voxelBytes, j->atlasDim, texPos )
                                  : fetch_nearest( j->atlas, j->opt.voxelBytes, j->atlasDim, texPos );
        float transferFn[4];
        float* const acc = ep->on ? ep->shadow : color;
        if( j->opt.tieBudget && !j->opt.filter && !ep->on ) /* test instrument, see orc_options */
            j->opt.tieBudget[pixelPos] += tie_budget( j, texPos, density, multiplyer, addedValue,
                                                      alphaCorrection, 1.0f - color[3], kStep, vpw );
        orc_tf_fetch( j->tf, density * multiplyer + addedValue, j->opt.tfFracBits, transferFn );
        orc_composite( transferFn, acc, alphaCorrection );
        if( !ep->on )
            ++*nSamples;
        if( j->opt.tieBudget )
        {
            if( ep->lo < 0.0f )
            {
                ep->eps = 5e-5f + 2.0f * j->opt.tieBudget[pixelPos];
                if( acc[3] > EARLY_EXIT - ep->eps )
                    ep->lo = acc[3];
            }
            if( ep->on )
            {
                if( acc[3] > EARLY_EXIT + ep->eps )
                    return 1;
                continue;
            }
            if( color[3] > EARLY_EXIT && color[3] <= EARLY_EXIT + ep->eps )
            {
                ep->on = 1; /* the march's own exit; an evaluation ep->eps lower goes on: follow it on a copy */
                ep->shadow[0] = color[0], ep->shadow[1] = color[1], ep->shadow[2] = color[2], ep->shadow[3] = color[3];
                continue;
            }
        }
        if( color[3] > EARLY_EXIT )
            return 1;
    }
    return 0;
}

static uint64_t raycast_pixel_ray_lod( const job_t* j, uint32_t x, uint32_t y )
{
    const orc_view_data* viewData = j->view;
    const lod_grid* g = j->lod;
    uint64_t nSamples = 0;
    ert_probe probe = { { 0, 0, 0, 0 }, -1.0f, 0.0f, 0 }; /* test instrument only */

    /* ray, global interval, clip planes, near plane: Renderer.cu:106-160 as in raycast_pixel */
    const f4 pixelEyeSpacePos =
        eye_space_from_window( (float)x, (float)y, viewData->glViewport, viewData->invProjMatrix );
    const f4 pixelWorldSpacePos = mat_mul_vec4( viewData->invViewMatrix, pixelEyeSpacePos );
    const f3 origin = { viewData->eyePosition[0], viewData->eyePosition[1], viewData->eyePosition[2] };
    const f3 d0 = { pixelWorldSpacePos.x - origin.x, pixelWorldSpacePos.y - origin.y,
                    pixelWorldSpacePos.z - origin.z };
    f3 dir = normalize_f3( d0 );
    if( dir.x == 0.0f ) dir.x = EPSILON;
    if( dir.y == 0.0f ) dir.y = EPSILON;
    if( dir.z == 0.0f ) dir.z = EPSILON;
    float tNearGlobal, tFarGlobal;
    const f3 globalBoxMin = { viewData->aabbMin[0], viewData->aabbMin[1], viewData->aabbMin[2] };
    const f3 globalBoxMax = { viewData->aabbMax[0], viewData->aabbMax[1], viewData->aabbMax[2] };
    if( !intersect_box( origin, dir, globalBoxMin, globalBoxMax, &tNearGlobal, &tFarGlobal ) )
        return 0;
    for( uint32_t i = 0; i < j->nPlanes; ++i )
    {
        const float* cp = j->clipPlanes + 4 * i;
        const f3 planeNormal = { cp[0], cp[1], cp[2] };
        float rn = dot3( dir, planeNormal );
        if( rn == 0.0f )
            rn = EPSILON;
        const float t = -( dot3( planeNormal, origin ) + cp[3] ) / rn;
        if( rn > 0.0f )
            tNearGlobal = fmaxf( tNearGlobal, t );
        else
            tFarGlobal = fminf( tFarGlobal, t );
    }
    if( tNearGlobal > tFarGlobal )
        return 0;
    float* px = j->pixelBuffer + ( (size_t)y * j->width + x ) * 4;
    if( px[3] > EARLY_EXIT )
        return 0;
    float color[4] = { px[0], px[1], px[2], px[3] };
    const f3 e3 = { pixelEyeSpacePos.x, pixelEyeSpacePos.y, pixelEyeSpacePos.z };
    const f3 nEye = normalize_f3( e3 );
    const float tNearPlane = -viewData->nearPlane / nEye.z;

    /* the ray's interval inside the box of all bricks */
    const f3 gridMin = { g->gmin[0], g->gmin[1], g->gmin[2] };
    const f3 gridMax = { g->gmax[0], g->gmax[1], g->gmax[2] };
    float t0, t1;
    const int any = intersect_box( origin, dir, gridMin, gridMax, &t0, &t1 );
    t0 = fmaxf( fmaxf( t0, tNearGlobal ), fmaxf( tNearPlane, 0.0f ) );
    t1 = fminf( t1, tFarGlobal );
    if( any && t0 < t1 )
    {
        const float o[3] = { origin.x, origin.y, origin.z };
        const float d[3] = { dir.x, dir.y, dir.z };
        const float invD[3] = { 1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z };
        const float tBase = tNearPlane * g->lodBase;
        const int maxHops = 3 * ( g->dim[0][0] + g->dim[0][1] + g->dim[0][2] ) + 16;
        float te = t0;
        for( int hop = 0; hop < maxHops && te < t1; ++hop )
        {
            int k = 0;
            float T = tBase;
            for( int lv = 1; lv < g->K; ++lv )
            {
                T = T + T;
                if( T <= te )
                    ++k;
            }
            const float tp = te + g->eps;
            const float p[3] = { o[0] + d[0] * tp, o[1] + d[1] * tp, o[2] + d[2] * tp };
            int node = -1;
            for( int s = 0; s < g->K && node < 0; ++s )
            {
                const int lv = s < g->K - k ? k + s : g->K - 1 - s; /* k..K-1, then k-1..0 */
                int c[3];
                for( int a = 0; a < 3; ++a )
                {
                    c[a] = (int)floorf( ( p[a] - g->gmin[a] ) * g->invCell[lv][a] );
                    if( c[a] < 0 ) c[a] = 0;
                    if( c[a] > g->dim[lv][a] - 1 ) c[a] = g->dim[lv][a] - 1;
                }
                node = g->tables[g->offset[lv] + ( (size_t)c[2] * g->dim[lv][1] + c[1] ) * g->dim[lv][0] + c[0]];
            }
            /* where the ray leaves the brick (no brick here: the finest level's grid cell) */
            float bmin[3], bmax[3];
            if( node >= 0 )
                for( int a = 0; a < 3; ++a )
                {
                    bmin[a] = j->nodes[node].aabbMin[a];
                    bmax[a] = j->nodes[node].aabbMin[a] + j->nodes[node].aabbSize[a];
                }
            else
                for( int a = 0; a < 3; ++a )
                {
                    int c = (int)floorf( ( p[a] - g->gmin[a] ) * g->invCell[0][a] );
                    if( c < 0 ) c = 0;
                    if( c > g->dim[0][a] - 1 ) c = g->dim[0][a] - 1;
                    bmin[a] = g->gmin[a] + g->cell[0][a] * (float)c;
                    bmax[a] = g->gmin[a] + g->cell[0][a] * (float)( c + 1 );
                }
            float tX = ( ( d[0] > 0.0f ? bmax[0] : bmin[0] ) - o[0] ) * invD[0];
            tX = fminf( tX, ( ( d[1] > 0.0f ? bmax[1] : bmin[1] ) - o[1] ) * invD[1] );
            tX = fminf( tX, ( ( d[2] > 0.0f ? bmax[2] : bmin[2] ) - o[2] ) * invD[2] );
            const float tB = fminf( fmaxf( tX, tp ), t1 ); /* always forward */
            if( node >= 0 &&
                integrate_run( j, &j->nodes[node], g->level[node], origin, dir, tp, tB, color, &nSamples,
                               (size_t)y * j->width + x, &probe ) )
                break;
            te = tB;
        }
    }
    if( j->opt.tieBudget && probe.lo >= 0.0f )
        j->opt.tieBudget[(size_t)y * j->width + x] += fmaxf( 0.0f, ( probe.on ? probe.shadow[3] : color[3] ) - probe.lo );
    px[0] = color[0];
    px[1] = color[1];
    px[2] = color[2];
    px[3] = color[3];
    return nSamples;
}

typedef struct
{
    const job_t* job;
    volatile uint32_t* nextRow;
    uint64_t samples;
} worker_t;

static void* worker_main( void* p )
{
    worker_t* w = (worker_t*)p;
    const job_t* j = w->job;
    uint64_t n = 0;
    for( ;; )
    {
        const uint32_t k = __sync_fetch_and_add( w->nextRow, 1u );
        const uint64_t y64 = (uint64_t)j->opt.rowBegin + (uint64_t)k * j->opt.rowStride;
        if( y64 >= j->opt.rowEnd || y64 >= j->height )
            break;
        const uint32_t y = (uint32_t)y64;
        for( uint32_t x = 0; x < j->width; ++x )
            n += j->lod ? raycast_pixel_ray_lod( j, x, y )
                        : j->opt.variant == 1 ? ( j->render->samplesPerPixel > 1u ? raycast_pixel_gl_ss( j, x, y ) : raycast_pixel_gl( j, x, y ) )
                                              : raycast_pixel( j, x, y );
    }
    w->samples = n;
    return NULL;
}

uint64_t orc_raycast( const uint8_t* atlas, const uint32_t atlasDim[3], float* pixelBuffer,
                      uint32_t width, uint32_t height, const float* clipPlanes,
                      uint32_t nPlanes, const float* tf, const orc_view_data* view,
                      uint32_t nodeCount, const orc_node_data* nodes,
                      const orc_render_data* render, const orc_options* optIn )
{
    job_t job;
    job.atlas = atlas;
    job.atlasDim = atlasDim;
    job.pixelBuffer = pixelBuffer;
    job.width = width;
    job.height = height;
    job.clipPlanes = clipPlanes;
    job.nPlanes = nPlanes;
    job.tf = tf;
    job.view = view;
    job.nodeCount = nodeCount;
    job.nodes = nodes;
    job.render = render;
    if( optIn )
        job.opt = *optIn;
    else
    {
        job.opt.tfFracBits = 8;
        job.opt.filter = 0;
        job.opt.nThreads = 1;
        job.opt.rowBegin = 0;
        job.opt.rowEnd = height;
        job.opt.rowStride = 1;
        job.opt.voxelBytes = 1;
        job.opt.variant = 0;
        job.opt.rayLod = 0;
        job.opt.lodScreenSpaceError = job.opt.lodWorldSpacePerPixel = 0.f;
        job.opt.tieBudget = NULL;
        job.opt.tieDelta = 0.f;
        job.opt.entryBias = 0.f;
    }
    job.lod = NULL;
    lod_grid* lodGrid = NULL;
    if( job.opt.rayLod )
    {
        if( job.opt.variant != 0 || !( job.opt.lodScreenSpaceError > 0.f ) ||
            !( job.opt.lodWorldSpacePerPixel > 0.f ) )
            return UINT64_MAX;
        lodGrid = lod_grid_build( nodes, nodeCount, atlasDim, job.opt.lodScreenSpaceError,
                                  job.opt.lodWorldSpacePerPixel );
        if( !lodGrid )
            return UINT64_MAX; /* the node list is not a brick hierarchy */
        job.lod = lodGrid;
    }
    if( job.opt.rowStride == 0 ) job.opt.rowStride = 1;
    if( job.opt.rowEnd == 0 || job.opt.rowEnd > height ) job.opt.rowEnd = height;
    int nt = job.opt.nThreads < 1 ? 1 : job.opt.nThreads;
    if( nt > 256 ) nt = 256;

    volatile uint32_t nextRow = 0;
    worker_t workers[256];
    pthread_t threads[256];
    for( int i = 0; i < nt; ++i )
    {
        workers[i].job = &job;
        workers[i].nextRow = &nextRow;
        workers[i].samples = 0;
    }
    if( nt == 1 )
        worker_main( &workers[0] );
    else
    {
        for( int i = 0; i < nt; ++i )
            pthread_create( &threads[i], NULL, worker_main, &workers[i] );
        for( int i = 0; i < nt; ++i )
            pthread_join( threads[i], NULL );
    }
    uint64_t total = 0;
    for( int i = 0; i < nt; ++i )
        total += workers[i].samples;
    lod_grid_free( lodGrid );
    return total;
}

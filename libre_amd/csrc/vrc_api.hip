/*
 * vrc_api.hip -- implementation of the C ABI declared in include/vrc_hip.h.
 *
 * Host-side logic of the device layer: context + stream, brick atlas (slot grid, free list,
 * pinned staging ring, upload stream + event ordering), classified-table cache, node table
 * and brick-grid construction, kernel selection, timing.  Replaces the host halves of
 * cuda::Renderer::Impl (cuda/Renderer.cu:232-333) and cuda::TexturePool (cuda/TexturePool.cu).
 */
#include "../../include/vrc_hip.h"
#include "vrc_internal.h"
#include "vrc_tables.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace
{
thread_local std::string g_lastError;

int fail( int code, const std::string& msg )
{
    g_lastError = msg;
    return code;
}

#define VRC_HIP_CHECK( expr )                                                              \
    do                                                                                     \
    {                                                                                      \
        const hipError_t _e = ( expr );                                                    \
        if( _e != hipSuccess )                                                             \
            return fail( VRC_EHIP, std::string( #expr ) + ": " + hipGetErrorString( _e ) ); \
    } while( 0 )

constexpr int kStagingSlots = 8;
constexpr int kNodeStages = 4;
} // namespace

struct vrc_pool
{
    vrc_ctx* ctx = nullptr; /* creating context (device identification only) */
    int device = 0;
    uint64_t uid = 0;       /* unique per process, keys the node-table caches of contexts */
    uint32_t elemBytes = 1;
    uint32_t maxBlock[3] = { 0, 0, 0 };
    uint32_t slotDim[3] = { 0, 0, 0 }; /* maxBlock rounded up to the micro-block size */
    uint32_t slots[3] = { 1, 1, 1 };
    uint32_t atlasDim[3] = { 0, 0, 0 };
    size_t slotBytes = 0, atlasBytes = 0;
    void* dAtlas = nullptr;
    bool bigAtlas = false; /* more than 2^32 voxels */
    /* tap-packed atlas of the trilinear filter (vrc_core.h): same slots, a texel of twice the voxel's bytes per voxel in
     * blocks of 64 rows of 9.
     * Allocated and filled from the byte atlas the first time a render asks for it (pool_enable_packed); from then
     * on every upload packs its slot too.  Guarded by `mutex`. */
    void* dPacked = nullptr;
    bool packedOn = false;
    bool packedFailed = false; /* the allocation was refused once: not tried again */

    std::mutex mutex; /* free list + staging ring index + upload event + render fences */
    /* slot position + "was in use before" flag: a released slot may still be read by a march
     * that is in flight; the upload that reuses it waits for the render fences first */
    std::vector< std::array< float, 4 > > freeList;
    struct RenderFence
    {
        const vrc_ctx* ctx;
        hipEvent_t event;
    };
    std::vector< RenderFence > renderFences; /* last march of every context that used this pool */

    hipStream_t uploadStream = nullptr; /* every write to the atlas (the repack kernels), in issue order */
    /* the host-to-device copies into the staging buffers alternate between two streams of their own: the DMA of
     * one brick runs while the repack kernel of the brick before it writes the atlas */
    hipStream_t copyStream[2] = { nullptr, nullptr };
    hipEvent_t lastUpload = nullptr;
    bool hasUpload = false;

    struct Staging
    {
        std::mutex mutex;
        void* pinned = nullptr;
        void* device = nullptr;
        hipEvent_t copied = nullptr; /* the brick is in `device` */
        hipEvent_t done = nullptr;   /* ... and has been repacked into the atlas: both buffers are free */
        bool used = false;
    };
    Staging staging[kStagingSlots];
    uint32_t nextStaging = 0;
};

struct vrc_ctx
{
    int device = 0;
    hipStream_t ownStream = nullptr;
    hipStream_t stream = nullptr;

    /* cuda::ColorMap + cuda::ClipPlanes state */
    float* dTf = nullptr;
    float* hTf = nullptr; /* pinned copy of the last uploaded TF: skips identical re-uploads */
    uint64_t tfVersion = 0;
    float planes[6][4];
    uint32_t nPlanes = 0;

    /* classified table */
    vrc_f4* dLut = nullptr;
    bool lutValid = false;
    uint64_t lutTfVersion = 0;
    vrc_lut_params lutParams = { 0, 0, 0, 0 };
    float lutMaxAlpha = 1.0f; /* largest opacity in the classified table (1: unknown) */
    bool tfGrey = false;      /* red == green == blue in every entry of the transfer function (bitwise) */
    int64_t optGreyTable = 1; /* VRC_OPT_GREY_TABLE */
    bool lutLinear = false;
    uint32_t lutLevels = 1;

    /* pixel buffer (cuda::PixelBufferObject) */
    vrc_f4* fbOwn = nullptr;
    size_t fbOwnPixels = 0;
    vrc_f4* fbExt = nullptr;
    uint32_t fbW = 0, fbH = 0;
    bool clearPending = false; /* pre_render's clear, folded into the first march or done lazily */

    /* node table + brick grid */
    vrc_dev_node* dNodes = nullptr;
    size_t dNodesCap = 0;
    int32_t* dGrid = nullptr;
    size_t dGridCap = 0;
    /* pinned staging for nodes + grid: a ring of kNodeStages areas of hStageCap bytes, each with
     * the event of the copies last issued from it, so a changed node list (a moving camera
     * re-sorts the bricks) does not have to wait for the frames still in the stream */
    void* hStage = nullptr;
    size_t hStageCap = 0;
    hipEvent_t stageEvent[4] = { nullptr, nullptr, nullptr, nullptr };
    bool stageUsed[4] = { false, false, false, false };
    uint32_t stageNext = 0;
    std::vector< vrc_node_data > cachedNodes;
    uint64_t cachedPoolUid = 0;
    bool cachedGridOk = false;
    bool cachedOneCell = false; /* every brick = one grid cell (vrc_host_tables::oneCellPerBrick) */
    bool cachedRayLod = false; /* tables are per-level (vrc_build_lod_tables) */
    bool cachedLodOk = false;
    uint32_t cachedLodLevels = 0;
    double cachedFinestVoxel = 0.0;
    bool cachedClamp = false;
    vrc_frame cachedGridFrame = {}; /* only grid* / lod* fields are meaningful */

    /* sort-first row map */
    uint32_t* dRowMap = nullptr;
    size_t dRowMapCap = 0;
    std::vector< uint32_t > rowMap;

    /* tile schedule */
    uint32_t* dTileOrder = nullptr;
    size_t dTileOrderCap = 0;
    bool tileOrderValid = false;
    uint32_t tileOrderReused = 0; /* frames in a row that kept the schedule of a nearby view */
    vrc_frame tileOrderFrame;
    int64_t optTileOrder = 1;
    int64_t optStepping = 1;
    int64_t optVariant = VRC_VARIANT_CUDARAYCASTER;
    bool rayLod = false; /* vrc_set_ray_lod */
    float rayLodSse = 1.0f, rayLodWorldPerPixel = 0.0f;

    unsigned long long* dCounter = nullptr;
    unsigned long long* hCounter = nullptr; /* pinned */

    /* one event pair per raycast launch since the last vrc_get_stats (ring, grows on demand) */
    std::vector< std::pair< hipEvent_t, hipEvent_t > > evPairs;
    size_t evUsed = 0;
    double evFoldedMs = 0.0;      /* pairs taken out of a full ring before anyone asked for the statistics */
    uint32_t evFoldedLaunches = 0;
    bool timed = false;

    int64_t optKernel = VRC_KERNEL_AUTO;
    int64_t optFilter = 0;
    int64_t optTfFracBits = 8;
    int64_t optCount = 0;
    int64_t optTiming = 1;
    int64_t optDepthSplit = 0;
    int64_t optErtParts = 0;     /* VRC_OPT_ERT_COMPACTION */
    int64_t optPackedAtlas = 1;  /* VRC_OPT_PACKED_ATLAS */
    uint32_t* dRayList = nullptr; /* counts | two ray lists (vrc_internal.h) */
    size_t dRayListCap = 0;       /* pixels */
    int lastErtParts = 0;         /* of the last vrc_render */

    vrc_stats stats = {};
};

int vrc_internal_fail( int code, const std::string& msg ) { return fail( code, msg ); }
static thread_local char tlsKernel[160] = "";
void vrc_internal_note_kernel( const char* fmt, ... )
{
    va_list ap;
    va_start( ap, fmt );
    vsnprintf( tlsKernel, sizeof( tlsKernel ), fmt, ap );
    va_end( ap );
}
static thread_local const void* tlsKernelFn = nullptr;
static thread_local int tlsKernelThreads = 0;
static thread_local size_t tlsKernelLds = 0;
void vrc_internal_note_kernel_fn( const void* fn, int threads, size_t dynamicLds )
{
    tlsKernelFn = fn;
    tlsKernelThreads = threads;
    tlsKernelLds = dynamicLds;
}
hipStream_t vrc_internal_ctx_stream( vrc_ctx* c, int* deviceOut )
{
    if( deviceOut )
        *deviceOut = c->device;
    return c->stream;
}

extern "C" {

const char* vrc_last_error( void ) { return g_lastError.c_str(); }
const char* vrc_last_kernel( void ) { return tlsKernel; }
int vrc_last_kernel_occupancy( int* workgroupsPerCu, int* threadsPerWorkgroup )
{
    if( !workgroupsPerCu )
        return fail( VRC_EINVAL, "vrc_last_kernel_occupancy: NULL argument" );
    *workgroupsPerCu = 0;
    if( threadsPerWorkgroup )
        *threadsPerWorkgroup = tlsKernelThreads;
    if( !tlsKernelFn )
        return fail( VRC_EINVAL, "vrc_last_kernel_occupancy: the calling thread's last vrc_render did not launch vrc_k_raycast" );
    int n = 0;
    VRC_HIP_CHECK( hipOccupancyMaxActiveBlocksPerMultiprocessor( &n, tlsKernelFn, tlsKernelThreads, tlsKernelLds ) );
    *workgroupsPerCu = n;
    return VRC_OK;
}
#if defined( VRC_DEV_BUILD )
int vrc_abi_version( void ) { return -VRC_ABI_VERSION; }
int vrc_is_dev_build( void ) { return 1; }
#else
int vrc_abi_version( void ) { return VRC_ABI_VERSION; }
int vrc_is_dev_build( void ) { return 0; }
#endif

/* ---------------------------------------------------------------------------------------- */
int vrc_ctx_create( int device, vrc_ctx** out )
{
    if( !out )
        return fail( VRC_EINVAL, "vrc_ctx_create: out is NULL" );
    *out = nullptr;
    int count = 0;
    VRC_HIP_CHECK( hipGetDeviceCount( &count ) );
    if( device < 0 || device >= count )
        return fail( VRC_EINVAL, "vrc_ctx_create: no such device" );
    VRC_HIP_CHECK( hipSetDevice( device ) );
    vrc_ctx* c = new vrc_ctx();
    c->device = device;
    std::memset( c->planes, 0, sizeof( c->planes ) );
    hipError_t e = hipStreamCreateWithFlags( &c->ownStream, hipStreamNonBlocking );
    if( e == hipSuccess ) e = hipMalloc( &c->dTf, 256 * 4 * sizeof( float ) );
    if( e == hipSuccess ) e = hipMalloc( &c->dLut, ( VRC_MAX_LOD_LEVELS * VRC_LUT_ENTRIES + 1u ) * sizeof( vrc_f4 ) );
    if( e == hipSuccess ) e = hipHostMalloc( &c->hTf, 256 * 4 * sizeof( float ) );
    if( e == hipSuccess ) e = hipMalloc( &c->dCounter, sizeof( unsigned long long ) );
    if( e == hipSuccess ) e = hipHostMalloc( &c->hCounter, sizeof( unsigned long long ) );
    if( e != hipSuccess )
    {
        const std::string msg = std::string( "vrc_ctx_create: " ) + hipGetErrorString( e );
        vrc_ctx_destroy( c );
        return fail( VRC_EHIP, msg );
    }
    c->stream = c->ownStream;
    *c->hCounter = 0;
    /* default transfer function: linear grey ramp (the reference uploads lexis' default
     * colour map in cuda::ColorMap::ColorMap, cuda/ColorMap.cu:32; that map lives in the
     * un-vendored Lexis library, so the default here is documented and explicit) */
    float tf[256 * 4];
    for( int i = 0; i < 256; ++i )
        tf[i * 4 + 0] = tf[i * 4 + 1] = tf[i * 4 + 2] = tf[i * 4 + 3] = (float)i / 255.0f;
    std::memcpy( c->hTf, tf, sizeof( tf ) );
    c->tfGrey = true;
    if( const char* g = std::getenv( "VRC_GREY_TABLE" ) ) /* default of VRC_OPT_GREY_TABLE for this process */
        c->optGreyTable = g[0] != '0';
    e = hipMemcpy( c->dTf, tf, sizeof( tf ), hipMemcpyHostToDevice );
    if( e != hipSuccess )
    {
        const std::string msg = std::string( "vrc_ctx_create: " ) + hipGetErrorString( e );
        vrc_ctx_destroy( c );
        return fail( VRC_EHIP, msg );
    }
    c->tfVersion = 1;
    *out = c;
    return VRC_OK;
}

void vrc_ctx_destroy( vrc_ctx* c )
{
    if( !c )
        return;
    (void)hipSetDevice( c->device );
    if( c->stream ) (void)hipStreamSynchronize( c->stream );
    if( c->dTf ) (void)hipFree( c->dTf );
    if( c->dLut ) (void)hipFree( c->dLut );
    if( c->hTf ) (void)hipHostFree( c->hTf );
    if( c->fbOwn ) (void)hipFree( c->fbOwn );
    if( c->dNodes ) (void)hipFree( c->dNodes );
    if( c->dGrid ) (void)hipFree( c->dGrid );
    if( c->hStage ) (void)hipHostFree( c->hStage );
    for( hipEvent_t e : c->stageEvent )
        if( e ) (void)hipEventDestroy( e );
    if( c->dTileOrder ) (void)hipFree( c->dTileOrder );
    if( c->dRayList ) (void)hipFree( c->dRayList );
    if( c->dRowMap ) (void)hipFree( c->dRowMap );
    if( c->dCounter ) (void)hipFree( c->dCounter );
    if( c->hCounter ) (void)hipHostFree( c->hCounter );
    for( auto& pr : c->evPairs )
    {
        (void)hipEventDestroy( pr.first );
        (void)hipEventDestroy( pr.second );
    }
    if( c->ownStream ) (void)hipStreamDestroy( c->ownStream );
    delete c;
}

int vrc_ctx_set_stream( vrc_ctx* c, void* s )
{
    if( !c )
        return fail( VRC_EINVAL, "vrc_ctx_set_stream: ctx is NULL" );
    VRC_HIP_CHECK( hipSetDevice( c->device ) );
    VRC_HIP_CHECK( hipStreamSynchronize( c->stream ) );
    c->stream = s ? (hipStream_t)s : c->ownStream;
    return VRC_OK;
}

int vrc_set_option( vrc_ctx* c, int option, int64_t value )
{
    if( !c )
        return fail( VRC_EINVAL, "vrc_set_option: ctx is NULL" );
    switch( option )
    {
    case VRC_OPT_KERNEL:
        if( value < VRC_KERNEL_AUTO || ( value > VRC_KERNEL_LDS && value != VRC_KERNEL_PACKED ) )
            return fail( VRC_EINVAL, "vrc_set_option: bad kernel variant" );
        c->optKernel = value;
        return VRC_OK;
    case VRC_OPT_FILTER:
        if( value != VRC_FILTER_NEAREST && value != VRC_FILTER_TRILINEAR )
            return fail( VRC_EINVAL, "vrc_set_option: filter is 0 (nearest) or 1 (trilinear)" );
        c->optFilter = value;
        return VRC_OK;
    case VRC_OPT_TF_FRAC_BITS:
        if( value < 0 || value > 16 )
            return fail( VRC_EINVAL, "vrc_set_option: tf frac bits out of range" );
        c->optTfFracBits = value;
        return VRC_OK;
    case VRC_OPT_COUNT_SAMPLES: c->optCount = value ? 1 : 0; return VRC_OK;
    case VRC_OPT_TILE_ORDER: c->optTileOrder = value ? 1 : 0; return VRC_OK;
    case VRC_OPT_STEPPING: c->optStepping = value ? 1 : 0; return VRC_OK;
    case VRC_OPT_KERNEL_TIMING: c->optTiming = value ? 1 : 0; return VRC_OK;
    case VRC_OPT_DEPTH_SPLIT: c->optDepthSplit = value ? 1 : 0; return VRC_OK;
    case VRC_OPT_GREY_TABLE: c->optGreyTable = value ? 1 : 0; return VRC_OK;
    case VRC_OPT_PACKED_ATLAS: c->optPackedAtlas = value ? 1 : 0; return VRC_OK;
    case VRC_OPT_ERT_COMPACTION:
        if( value < 0 || value > VRC_MAX_ERT_PARTS )
            return fail( VRC_EINVAL, "VRC_OPT_ERT_COMPACTION: 0 (off) or 2.." + std::to_string( VRC_MAX_ERT_PARTS ) +
                                         " launches per frame" );
        c->optErtParts = value < 2 ? 0 : value;
        return VRC_OK;
    case VRC_OPT_VARIANT:
        if( value != VRC_VARIANT_CUDARAYCASTER && value != VRC_VARIANT_GLRAYCASTER )
            return fail( VRC_EINVAL, "vrc_set_option: variant is 0 (cudaRaycaster) or 1 (glRaycaster)" );
        c->optVariant = value;
        return VRC_OK;
    default: return fail( VRC_EINVAL, "vrc_set_option: unknown option" );
    }
}

int vrc_get_option( vrc_ctx* c, int option, int64_t* value )
{
    if( !c || !value )
        return fail( VRC_EINVAL, "vrc_get_option: NULL argument" );
    switch( option )
    {
    case VRC_OPT_KERNEL: *value = c->optKernel; return VRC_OK;
    case VRC_OPT_FILTER: *value = c->optFilter; return VRC_OK;
    case VRC_OPT_TF_FRAC_BITS: *value = c->optTfFracBits; return VRC_OK;
    case VRC_OPT_COUNT_SAMPLES: *value = c->optCount; return VRC_OK;
    case VRC_OPT_TILE_ORDER: *value = c->optTileOrder; return VRC_OK;
    case VRC_OPT_STEPPING: *value = c->optStepping; return VRC_OK;
    case VRC_OPT_KERNEL_TIMING: *value = c->optTiming; return VRC_OK;
    case VRC_OPT_DEPTH_SPLIT: *value = c->optDepthSplit; return VRC_OK;
    case VRC_OPT_ERT_COMPACTION: *value = c->optErtParts; return VRC_OK;
    case VRC_OPT_GREY_TABLE: *value = c->optGreyTable; return VRC_OK;
    case VRC_OPT_PACKED_ATLAS: *value = c->optPackedAtlas; return VRC_OK;
    case VRC_OPT_VARIANT: *value = c->optVariant; return VRC_OK;
    case VRC_OPT_KERNEL_USED: *value = c->stats.kernel_variant; return VRC_OK;
    default: return fail( VRC_EINVAL, "vrc_get_option: unknown option" );
    }
}

/* per-ray adaptive LOD (extension): see include/vrc_hip.h */
int vrc_set_ray_lod( vrc_ctx* c, int enable, float screenSpaceError, float worldSpacePerPixel )
{
    if( !c )
        return fail( VRC_EINVAL, "vrc_set_ray_lod: ctx is NULL" );
    if( enable && ( !( screenSpaceError > 0.0f ) || !( worldSpacePerPixel > 0.0f ) ) )
        return fail( VRC_EINVAL, "vrc_set_ray_lod: screen-space error and world space per pixel must be > 0" );
    c->rayLod = enable != 0;
    if( enable )
    {
        c->rayLodSse = screenSpaceError;
        c->rayLodWorldPerPixel = worldSpacePerPixel;
    }
    return VRC_OK;
}

/* ---------------------------------------------------------------------------------------- */
int vrc_pool_create( vrc_ctx* c, size_t bytesPerVoxel, int isSigned, int isFloat,
                     size_t nComponents, const uint32_t maxBlock[3], size_t maxBytes,
                     vrc_pool** out )
{
    if( !c || !out || !maxBlock )
        return fail( VRC_EINVAL, "vrc_pool_create: NULL argument" );
    *out = nullptr;
    /* cuda/TexturePool.cu:66-67 */
    if( nComponents == 0 || nComponents > 4 )
        return fail( VRC_EUNSUPPORTED, "Channel number cannot be 0 or larger than 4" );
    /* the reference kernel only ever fetches unsigned char (Renderer.cu:211, quirk Q2); this
     * layer renders unsigned 8-bit (the reference path) and, as an extension, unsigned 16-bit
     * single-channel volumes, and says so for the rest instead of mis-rendering them */
    if( ( bytesPerVoxel != 1 && bytesPerVoxel != 2 ) || nComponents != 1 || isFloat || isSigned )
        return fail( VRC_EUNSUPPORTED,
                     "vrc_pool_create: only unsigned 8/16-bit single-channel volumes are implemented" );
    if( maxBlock[0] == 0 || maxBlock[1] == 0 || maxBlock[2] == 0 )
        return fail( VRC_EINVAL, "vrc_pool_create: zero block size" );
    VRC_HIP_CHECK( hipSetDevice( c->device ) );

    static std::mutex uidMutex;
    static uint64_t nextUid = 1;
    vrc_pool* p = new vrc_pool();
    p->ctx = c;
    p->device = c->device;
    {
        std::lock_guard< std::mutex > lock( uidMutex );
        p->uid = nextUid++;
    }
    p->elemBytes = (uint32_t)( bytesPerVoxel * nComponents );
    for( int a = 0; a < 3; ++a )
    {
        p->maxBlock[a] = maxBlock[a];
        p->slotDim[a] = ( maxBlock[a] + VRC_MB - 1 ) / VRC_MB * VRC_MB;
        if( p->slotDim[a] > VRC_MAX_TEXTURE_3D )
        {
            delete p;
            return fail( VRC_EINVAL, "vrc_pool_create: block larger than VRC_MAX_TEXTURE_3D" );
        }
    }
    p->slotBytes = (size_t)p->slotDim[0] * p->slotDim[1] * p->slotDim[2] * p->elemBytes;
    if( (size_t)p->slotDim[0] * p->slotDim[1] * p->slotDim[2] >= ( 1u << 24 ) )
    {
        delete p;
        return fail( VRC_EINVAL, "vrc_pool_create: block of 2^24 voxels or more (max 248^3)" );
    }

    /* cuda/TexturePool.cu:119-135, with 64-bit arithmetic (fixes quirk Q12) */
    size_t freeMem = 0, totalMem = 0;
    hipError_t e = hipMemGetInfo( &freeMem, &totalMem );
    if( e != hipSuccess )
    {
        delete p;
        return fail( VRC_EHIP, std::string( "hipMemGetInfo: " ) + hipGetErrorString( e ) );
    }
    const size_t maxMemory = std::min( freeMem, maxBytes );
    const uint64_t maxBlocks64 = maxMemory / p->slotBytes;
    const uint32_t maxBlocks = (uint32_t)std::min< uint64_t >( maxBlocks64, 0xFFFFFFFFull );
    p->slots[0] = std::min( VRC_MAX_TEXTURE_3D / p->slotDim[0], std::max( maxBlocks, 1u ) );
    p->slots[1] = std::min( VRC_MAX_TEXTURE_3D / p->slotDim[1],
                            std::max( maxBlocks / p->slots[0], 1u ) );
    p->slots[2] = std::min( VRC_MAX_TEXTURE_3D / p->slotDim[2],
                            std::max( maxBlocks / ( p->slots[0] * p->slots[1] ), 1u ) );
    for( int a = 0; a < 3; ++a )
        p->atlasDim[a] = p->slots[a] * p->slotDim[a];
    p->atlasBytes = (size_t)p->slots[0] * p->slots[1] * p->slots[2] *
                    vrc_slot_elems( p->slotDim[0], p->slotDim[1], p->slotDim[2] ) * p->elemBytes;
    /* more than 2^32 voxels: slot bases become 64-bit (BIG kernel instances; the LDS-staged and the
     * per-ray LOD kernels, whose slot bases are 32-bit, are not offered for such a pool) */
    p->bigAtlas = p->atlasBytes / p->elemBytes >= 0xFFFFFFFFull;

    /* cuda/TexturePool.cu:137-144: i,j,k descending, k innermost; slots are popped from the back */
    for( int i = (int)p->slots[0] - 1; i >= 0; --i )
        for( int j = (int)p->slots[1] - 1; j >= 0; --j )
            for( int k = (int)p->slots[2] - 1; k >= 0; --k )
                p->freeList.push_back( { (float)i / (float)p->slots[0],
                                         (float)j / (float)p->slots[1],
                                         (float)k / (float)p->slots[2], 0.0f } );

    e = hipMalloc( &p->dAtlas, p->atlasBytes );
    if( e == hipSuccess ) e = hipStreamCreateWithFlags( &p->uploadStream, hipStreamNonBlocking );
    /* the clear goes on the upload stream, where every write to the atlas is queued: a hipMemset on the null stream is
     * not ordered with a non-blocking stream, and the first uploads of a large pool could be overtaken by it (seen once:
     * a 6 GB pool of 16-bit voxels, round 4) */
    if( e == hipSuccess ) e = hipMemsetAsync( p->dAtlas, 0, p->atlasBytes, p->uploadStream );
    if( e == hipSuccess ) e = hipEventCreateWithFlags( &p->lastUpload, hipEventDisableTiming );
    if( e == hipSuccess )
    {
        e = hipEventRecord( p->lastUpload, p->uploadStream ); /* a render before any upload waits for the clear too */
        p->hasUpload = true;
    }
    for( int k = 0; k < 2 && e == hipSuccess; ++k )
        e = hipStreamCreateWithFlags( &p->copyStream[k], hipStreamNonBlocking );
    for( int s = 0; s < kStagingSlots && e == hipSuccess; ++s )
    {
        e = hipHostMalloc( &p->staging[s].pinned, p->slotBytes );
        if( e == hipSuccess ) e = hipMalloc( &p->staging[s].device, p->slotBytes );
        if( e == hipSuccess )
            e = hipEventCreateWithFlags( &p->staging[s].copied, hipEventDisableTiming );
        if( e == hipSuccess )
            e = hipEventCreateWithFlags( &p->staging[s].done, hipEventDisableTiming );
    }
    if( e != hipSuccess )
    {
        const std::string msg = std::string( "vrc_pool_create: " ) + hipGetErrorString( e );
        vrc_pool_destroy( p );
        return fail( e == hipErrorOutOfMemory ? VRC_ENOMEM : VRC_EHIP, msg );
    }
    *out = p;
    return VRC_OK;
}

void vrc_pool_destroy( vrc_pool* p )
{
    if( !p )
        return;
    /* the caller guarantees no context is still rendering from this pool */
    (void)hipSetDevice( p->device );
    (void)hipDeviceSynchronize();
    for( int s = 0; s < kStagingSlots; ++s )
    {
        if( p->staging[s].pinned ) (void)hipHostFree( p->staging[s].pinned );
        if( p->staging[s].device ) (void)hipFree( p->staging[s].device );
        if( p->staging[s].copied ) (void)hipEventDestroy( p->staging[s].copied );
        if( p->staging[s].done ) (void)hipEventDestroy( p->staging[s].done );
    }
    for( int k = 0; k < 2; ++k )
        if( p->copyStream[k] ) (void)hipStreamDestroy( p->copyStream[k] );
    if( p->lastUpload ) (void)hipEventDestroy( p->lastUpload );
    for( auto& f : p->renderFences )
        (void)hipEventDestroy( f.event );
    if( p->uploadStream ) (void)hipStreamDestroy( p->uploadStream );
    if( p->dPacked ) (void)hipFree( p->dPacked );
    if( p->dAtlas ) (void)hipFree( p->dAtlas );
    delete p;
}

/* fences: the render events a reused slot's upload has to wait for (empty for a fresh slot) */
static int pool_take_slot( vrc_pool* p, float slot[3], std::vector< hipEvent_t >& fences )
{
    std::lock_guard< std::mutex > lock( p->mutex );
    if( p->freeList.empty() )
    {
        slot[0] = slot[1] = slot[2] = -1.0f; /* INVALID_SLOT_POSITION, TexturePool.cu:98 */
        return fail( VRC_EFULL, "vrc_pool_copy_to_slot: no free slot" );
    }
    const auto s = p->freeList.back();
    p->freeList.pop_back();
    slot[0] = s[0];
    slot[1] = s[1];
    slot[2] = s[2];
    if( s[3] != 0.0f )
        for( const auto& f : p->renderFences )
            fences.push_back( f.event );
    return VRC_OK;
}

static void pool_slot_voxel( const vrc_pool* p, const float slot[3], uint32_t o[3] )
{
    /* cuda/TexturePool.cu:193-197 */
    for( int a = 0; a < 3; ++a )
        o[a] = (uint32_t)std::lround( slot[a] * (float)p->atlasDim[a] );
}

static int pool_check_size( const vrc_pool* p, const uint32_t size[3] )
{
    for( int a = 0; a < 3; ++a )
        if( size[a] == 0 || size[a] > p->slotDim[a] )
            return fail( VRC_EINVAL, "vrc_pool_copy_to_slot: brick does not fit a slot" );
    return VRC_OK;
}

static int pool_upload( vrc_pool* p, const void* src, bool srcIsDevice, const uint32_t size[3],
                        float slotOut[3] )
{
    if( !p || !src || !size || !slotOut )
        return fail( VRC_EINVAL, "vrc_pool_copy_to_slot: NULL argument" );
    slotOut[0] = slotOut[1] = slotOut[2] = -1.0f;
    int rc = pool_check_size( p, size );
    if( rc != VRC_OK )
        return rc;
    VRC_HIP_CHECK( hipSetDevice( p->device ) );
    float slot[3];
    std::vector< hipEvent_t > fences;
    rc = pool_take_slot( p, slot, fences );
    if( rc != VRC_OK )
        return rc;
    uint32_t o[3];
    pool_slot_voxel( p, slot, o );
    const size_t bytes = (size_t)size[0] * size[1] * size[2] * p->elemBytes;

    uint32_t si;
    {
        std::lock_guard< std::mutex > lock( p->mutex );
        si = p->nextStaging++ % kStagingSlots;
    }
    vrc_pool::Staging& st = p->staging[si];
    hipError_t e = hipSuccess;
    {
        std::lock_guard< std::mutex > slock( st.mutex );
        if( st.used )
            e = hipEventSynchronize( st.done ); /* staging buffers free again */
        /* a slot that was released may still be read by a march in flight (the texture cache
         * drops an object as soon as the host is done with it): order the overwrite after the
         * last march of every context that rendered from this pool */
        for( size_t i = 0; i < fences.size() && e == hipSuccess; ++i )
            e = hipStreamWaitEvent( p->uploadStream, fences[i], 0 );
        const void* devSrc = src;
        if( e == hipSuccess && !srcIsDevice )
        {
            std::memcpy( st.pinned, src, bytes ); /* the host pointer is only borrowed */
            hipStream_t const cs = p->copyStream[si & 1u];
            e = hipMemcpyAsync( st.device, st.pinned, bytes, hipMemcpyHostToDevice, cs );
            if( e == hipSuccess ) e = hipEventRecord( st.copied, cs );
            if( e == hipSuccess ) e = hipStreamWaitEvent( p->uploadStream, st.copied, 0 );
            devSrc = st.device;
        }
        if( e == hipSuccess )
        {
            vrc_layout lay;
            for( int a = 0; a < 3; ++a )
            {
                lay.slots[a] = p->slots[a];
                lay.slotDim[a] = p->slotDim[a];
            }
            const uint64_t base = vrc_slot_base( lay, o[0] / p->slotDim[0], o[1] / p->slotDim[1],
                                                 o[2] / p->slotDim[2] );
            uint8_t* const slotPtr = (uint8_t*)p->dAtlas + (size_t)base * p->elemBytes;
            /* the writes of one upload are queued as a unit (pool_enable_packed packs every slot written before it
             * was called; every upload after it packs its own) */
            std::lock_guard< std::mutex > lock( p->mutex );
            e = vrc_launch_repack_brick( devSrc, slotPtr, p->elemBytes, size, p->slotDim, p->uploadStream );
            if( e == hipSuccess && p->packedOn )
                e = vrc_launch_pack_slots( p->dAtlas, p->dPacked, base,
                                           (uint64_t)p->slotDim[0] * p->slotDim[1] * p->slotDim[2], p->slotDim,
                                           p->elemBytes, p->uploadStream );
            if( e == hipSuccess )
                e = hipEventRecord( st.done, p->uploadStream );
            if( e == hipSuccess )
            {
                e = hipEventRecord( p->lastUpload, p->uploadStream );
                p->hasUpload = true;
            }
        }
        st.used = true;
    }
    if( e != hipSuccess )
    {
        vrc_pool_release_slot( p, slot );
        return fail( VRC_EHIP, std::string( "vrc_pool_copy_to_slot: " ) + hipGetErrorString( e ) );
    }
    if( srcIsDevice )
    {
        /* the caller's device buffer must stay valid until the repack has run */
        e = hipStreamSynchronize( p->uploadStream );
        if( e != hipSuccess )
            return fail( VRC_EHIP, std::string( "vrc_pool_copy_to_slot_device: " ) +
                                       hipGetErrorString( e ) );
    }
    slotOut[0] = slot[0];
    slotOut[1] = slot[1];
    slotOut[2] = slot[2];
    return VRC_OK;
}

/* The tap-packed atlas of the trilinear filter, on first use: 2.25 times the atlas's bytes, filled from it on the upload
 * stream behind every upload queued so far.  false (and no error) when the pool cannot have one: voxels of more than
 * 16 bits, or not enough device memory (tried once). */
static bool pool_packed_possible( const vrc_pool* p )
{
    return ( p->elemBytes == 1 || p->elemBytes == 2 ) && VRC_LAYOUT == 0;
}
/* bytes of the pool's packed atlas (+ 8: a pair is read as 4 / 8 bytes at the last texel) */
static uint64_t pool_packed_bytes( const vrc_pool* p )
{
    return vrc_packed_elems( p->atlasBytes / p->elemBytes ) * VRC_PK_TEXEL( p->elemBytes ) + 8u;
}
static bool pool_enable_packed( vrc_pool* p )
{
    std::lock_guard< std::mutex > lock( p->mutex );
    if( p->packedOn )
        return true;
    if( p->packedFailed || !pool_packed_possible( p ) )
        return false;
    const size_t bytes = (size_t)pool_packed_bytes( p );
    size_t freeMem = 0, totalMem = 0;
    hipError_t e = hipMemGetInfo( &freeMem, &totalMem );
    /* leave a margin for the caller's frame buffers and staging */
    if( e == hipSuccess && bytes + ( 256u << 20 ) > freeMem )
        e = hipErrorOutOfMemory;
    if( e == hipSuccess )
        e = hipMalloc( &p->dPacked, bytes );
    if( e == hipSuccess )
        e = vrc_launch_pack_slots( p->dAtlas, p->dPacked, 0u, p->atlasBytes / p->elemBytes, p->slotDim, p->elemBytes,
                                   p->uploadStream );
    if( e == hipSuccess )
    {
        e = hipEventRecord( p->lastUpload, p->uploadStream );
        p->hasUpload = true;
    }
    if( e != hipSuccess )
    {
        (void)hipGetLastError();
        if( p->dPacked )
        {
            (void)hipStreamSynchronize( p->uploadStream );
            (void)hipFree( p->dPacked );
        }
        p->dPacked = nullptr;
        p->packedFailed = true;
        return false;
    }
    p->packedOn = true;
    return true;
}

int vrc_pool_copy_to_slot( vrc_pool* p, const void* hostBrick, const uint32_t size[3],
                           float slotOut[3] )
{
    return pool_upload( p, hostBrick, false, size, slotOut );
}

int vrc_pool_copy_to_slot_device( vrc_pool* p, const void* deviceBrick, const uint32_t size[3],
                                  float slotOut[3] )
{
    return pool_upload( p, deviceBrick, true, size, slotOut );
}

int vrc_pool_release_slot( vrc_pool* p, const float slot[3] )
{
    if( !p || !slot )
        return fail( VRC_EINVAL, "vrc_pool_release_slot: NULL argument" );
    if( slot[0] < 0.f || slot[1] < 0.f || slot[2] < 0.f )
        return fail( VRC_EINVAL, "vrc_pool_release_slot: invalid slot" );
    std::lock_guard< std::mutex > lock( p->mutex );
    p->freeList.push_back( { slot[0], slot[1], slot[2], 1.0f } );
    return VRC_OK;
}

int vrc_pool_info( const vrc_pool* p, size_t* slotBytes, uint32_t atlasDim[3], size_t* atlasBytes,
                   uint32_t slots[3], uint32_t* freeSlots )
{
    if( !p )
        return fail( VRC_EINVAL, "vrc_pool_info: pool is NULL" );
    if( slotBytes ) *slotBytes = p->slotBytes;
    if( atlasBytes ) *atlasBytes = p->atlasBytes;
    for( int a = 0; a < 3; ++a )
    {
        if( atlasDim ) atlasDim[a] = p->atlasDim[a];
        if( slots ) slots[a] = p->slots[a];
    }
    if( freeSlots )
    {
        std::lock_guard< std::mutex > lock( const_cast< vrc_pool* >( p )->mutex );
        *freeSlots = (uint32_t)p->freeList.size();
    }
    return VRC_OK;
}

int vrc_pool_synchronize( vrc_pool* p )
{
    if( !p )
        return fail( VRC_EINVAL, "vrc_pool_synchronize: pool is NULL" );
    VRC_HIP_CHECK( hipSetDevice( p->device ) );
    VRC_HIP_CHECK( hipStreamSynchronize( p->uploadStream ) );
    return VRC_OK;
}

int vrc_pool_read_region( vrc_pool* p, const uint32_t origin[3], const uint32_t size[3],
                          void* hostOut )
{
    if( !p || !origin || !size || !hostOut )
        return fail( VRC_EINVAL, "vrc_pool_read_region: NULL argument" );
    for( int a = 0; a < 3; ++a )
        if( (uint64_t)origin[a] + size[a] > p->atlasDim[a] )
            return fail( VRC_EINVAL, "vrc_pool_read_region: region outside the atlas" );
    VRC_HIP_CHECK( hipSetDevice( p->device ) );
    const size_t bytes = (size_t)size[0] * size[1] * size[2] * p->elemBytes;
    if( bytes == 0 )
        return VRC_OK;
    void* tmp = nullptr;
    VRC_HIP_CHECK( hipMalloc( &tmp, bytes ) );
    hipError_t e = hipStreamSynchronize( p->uploadStream );
    if( e == hipSuccess )
    {
        vrc_layout lay;
        for( int a = 0; a < 3; ++a )
        {
            lay.slots[a] = p->slots[a];
            lay.slotDim[a] = p->slotDim[a];
        }
        e = vrc_launch_read_region( p->dAtlas, tmp, p->elemBytes, origin, size, lay,
                                    p->uploadStream );
    }
    if( e == hipSuccess ) e = hipStreamSynchronize( p->uploadStream );
    if( e == hipSuccess ) e = hipMemcpy( hostOut, tmp, bytes, hipMemcpyDeviceToHost );
    (void)hipFree( tmp );
    if( e != hipSuccess )
        return fail( VRC_EHIP, std::string( "vrc_pool_read_region: " ) + hipGetErrorString( e ) );
    return VRC_OK;
}

int vrc_pool_histogram( vrc_pool* p, const float slot[3], const uint32_t origin[3],
                        const uint32_t size[3], uint32_t binCount, uint64_t scaleFactor,
                        uint64_t* hostBins )
{
    if( !p || !slot || !origin || !size || !hostBins )
        return fail( VRC_EINVAL, "vrc_pool_histogram: NULL argument" );
    if( slot[0] < 0.f || slot[1] < 0.f || slot[2] < 0.f )
        return fail( VRC_EINVAL, "vrc_pool_histogram: invalid slot" );
    const uint32_t typeRange = p->elemBytes == 1 ? 256u : 65536u;
    if( binCount == 0 || binCount > 4096 || typeRange % binCount != 0 )
        return fail( VRC_EINVAL, "vrc_pool_histogram: bin count must divide the voxel type's range (max 4096)" );
    for( int a = 0; a < 3; ++a )
        if( (uint64_t)origin[a] + size[a] > p->slotDim[a] )
            return fail( VRC_EINVAL, "vrc_pool_histogram: region outside the slot" );
    VRC_HIP_CHECK( hipSetDevice( p->device ) );
    uint32_t o[3];
    pool_slot_voxel( p, slot, o );
    vrc_layout lay;
    for( int a = 0; a < 3; ++a )
    {
        lay.slots[a] = p->slots[a];
        lay.slotDim[a] = p->slotDim[a];
    }
    const uint64_t base = vrc_slot_base( lay, o[0] / p->slotDim[0], o[1] / p->slotDim[1], o[2] / p->slotDim[2] );
    unsigned long long* dBins = nullptr;
    VRC_HIP_CHECK( hipMalloc( &dBins, binCount * sizeof( unsigned long long ) ) );
    /* on the upload stream: ordered after the brick's own upload */
    hipError_t e = hipMemsetAsync( dBins, 0, binCount * sizeof( unsigned long long ), p->uploadStream );
    if( e == hipSuccess )
        e = vrc_launch_brick_histogram( (const uint8_t*)p->dAtlas + (size_t)base * p->elemBytes, p->elemBytes,
                                        p->slotDim[0] / VRC_MB, p->slotDim[1] / VRC_MB, origin, size,
                                        binCount, (unsigned long long)scaleFactor, dBins, p->uploadStream );
    if( e == hipSuccess )
        e = hipMemcpyAsync( hostBins, dBins, binCount * sizeof( unsigned long long ), hipMemcpyDeviceToHost,
                            p->uploadStream );
    if( e == hipSuccess )
        e = hipStreamSynchronize( p->uploadStream );
    (void)hipFree( dBins );
    if( e != hipSuccess )
        return fail( VRC_EHIP, std::string( "vrc_pool_histogram: " ) + hipGetErrorString( e ) );
    return VRC_OK;
}

/* ---------------------------------------------------------------------------------------- */
int vrc_update( vrc_ctx* c, const float tf[256 * 4], const float* planes, uint32_t nPlanes )
{
    if( !c )
        return fail( VRC_EINVAL, "vrc_update: ctx is NULL" );
    if( nPlanes > 6 )
        return fail( VRC_EINVAL, "vrc_update: more than 6 clip planes" );
    if( nPlanes > 0 && !planes )
        return fail( VRC_EINVAL, "vrc_update: planes is NULL" );
    VRC_HIP_CHECK( hipSetDevice( c->device ) );
    if( tf && std::memcmp( c->hTf, tf, 256 * 4 * sizeof( float ) ) != 0 )
    {
        /* 4 KiB, as cudaMemcpyToArray in cuda/ColorMap.cu:61-64.  The reference re-uploads every
         * frame (CudaRaycastRenderer.cpp:109-110); an unchanged map is skipped here, so the
         * per-frame call neither copies nor synchronizes.  The pinned copy may only be
         * rewritten once the previous async copy from it has been consumed. */
        VRC_HIP_CHECK( hipStreamSynchronize( c->stream ) );
        std::memcpy( c->hTf, tf, 256 * 4 * sizeof( float ) );
        c->tfGrey = true;
        for( int i = 0; i < 256 && c->tfGrey; ++i )
            c->tfGrey = std::memcmp( c->hTf + 4 * i, c->hTf + 4 * i + 1, sizeof( float ) ) == 0 &&
                        std::memcmp( c->hTf + 4 * i, c->hTf + 4 * i + 2, sizeof( float ) ) == 0;
        VRC_HIP_CHECK( hipMemcpyAsync( c->dTf, c->hTf, 256 * 4 * sizeof( float ),
                                       hipMemcpyHostToDevice, c->stream ) );
        ++c->tfVersion;
    }
    c->nPlanes = nPlanes;
    for( uint32_t i = 0; i < nPlanes; ++i )
        for( int k = 0; k < 4; ++k )
            c->planes[i][k] = planes[i * 4 + k];
    return VRC_OK;
}

static vrc_f4* ctx_fb( vrc_ctx* c ) { return c->fbExt ? c->fbExt : c->fbOwn; }

int vrc_set_row_map( vrc_ctx* c, const uint32_t* rows, uint32_t n )
{
    if( !c || ( n && !rows ) )
        return fail( VRC_EINVAL, "vrc_set_row_map: NULL argument" );
    VRC_HIP_CHECK( hipSetDevice( c->device ) );
    if( n == c->rowMap.size() && ( n == 0 || std::memcmp( rows, c->rowMap.data(), n * sizeof( uint32_t ) ) == 0 ) )
        return VRC_OK;
    VRC_HIP_CHECK( hipStreamSynchronize( c->stream ) );
    if( n > c->dRowMapCap )
    {
        if( c->dRowMap ) VRC_HIP_CHECK( hipFree( c->dRowMap ) );
        c->dRowMap = nullptr;
        c->dRowMapCap = 0;
        VRC_HIP_CHECK( hipMalloc( &c->dRowMap, n * sizeof( uint32_t ) ) );
        c->dRowMapCap = n;
    }
    if( n )
        VRC_HIP_CHECK( hipMemcpy( c->dRowMap, rows, n * sizeof( uint32_t ), hipMemcpyHostToDevice ) );
    c->rowMap.assign( rows, rows + n );
    c->tileOrderValid = false;
    return VRC_OK;
}

int vrc_pre_render( vrc_ctx* c, const vrc_view_data* view )
{
    if( !c || !view )
        return fail( VRC_EINVAL, "vrc_pre_render: NULL argument" );
    const uint32_t w = view->glViewport[2];
    const uint32_t h = c->rowMap.empty() ? view->glViewport[3] : (uint32_t)c->rowMap.size();
    if( w == 0 || h == 0 || view->glViewport[3] == 0 )
        return fail( VRC_EINVAL, "vrc_pre_render: empty viewport" );
    /* pixels are indexed in 32 bits (y * width + x) */
    if( (uint64_t)w * h > 0xFFFFFFFFull )
        return fail( VRC_EINVAL, "vrc_pre_render: a pixel buffer of 2^32 pixels or more" );
    for( uint32_t r : c->rowMap )
        if( r >= view->glViewport[3] )
            return fail( VRC_EINVAL, "vrc_pre_render: row map entry outside the frame" );
    VRC_HIP_CHECK( hipSetDevice( c->device ) );
    if( c->fbExt )
    {
        if( w != c->fbW || h != c->fbH )
            return fail( VRC_EINVAL, "vrc_pre_render: viewport differs from the external framebuffer" );
    }
    else
    {
        const size_t pixels = (size_t)w * h;
        if( pixels > c->fbOwnPixels )
        {
            VRC_HIP_CHECK( hipStreamSynchronize( c->stream ) );
            if( c->fbOwn ) VRC_HIP_CHECK( hipFree( c->fbOwn ) );
            c->fbOwn = nullptr;
            c->fbOwnPixels = 0;
            VRC_HIP_CHECK( hipMalloc( &c->fbOwn, pixels * sizeof( vrc_f4 ) ) );
            c->fbOwnPixels = pixels;
        }
        c->fbW = w;
        c->fbH = h;
    }
    /* PixelBufferObject::mapBuffer clears the mapped buffer (cuda/PixelBufferObject.cu:80).  The
     * clear is folded into the first march of the frame (vrc_frame::clearFirst); whoever looks at
     * the buffer before a march has run gets it done then (resolve_clear). */
    c->clearPending = true;
    return VRC_OK;
}

/* the pixel buffer is about to be read or re-pointed: do the clear pre_render promised */
static int resolve_clear( vrc_ctx* c )
{
    if( c->clearPending && ctx_fb( c ) && c->fbW && c->fbH )
    {
        VRC_HIP_CHECK( hipSetDevice( c->device ) );
        VRC_HIP_CHECK( hipMemsetAsync( ctx_fb( c ), 0, (size_t)c->fbW * c->fbH * sizeof( vrc_f4 ), c->stream ) );
    }
    c->clearPending = false;
    return VRC_OK;
}

int vrc_set_framebuffer( vrc_ctx* c, void* deviceRgba, uint32_t width, uint32_t height )
{
    if( !c )
        return fail( VRC_EINVAL, "vrc_set_framebuffer: ctx is NULL" );
    if( deviceRgba && ( width == 0 || height == 0 ) )
        return fail( VRC_EINVAL, "vrc_set_framebuffer: empty framebuffer" );
    if( deviceRgba && ( (uintptr_t)deviceRgba % 16u ) != 0 )
        return fail( VRC_EINVAL, "vrc_set_framebuffer: pointer must be 16-byte aligned" );
    {
        const int rc = resolve_clear( c );
        if( rc != VRC_OK )
            return rc;
    }
    c->fbExt = (vrc_f4*)deviceRgba;
    if( deviceRgba )
    {
        c->fbW = width;
        c->fbH = height;
    }
    else
    {
        c->fbW = c->fbH = 0;
        if( c->fbOwn )
        {
            (void)hipSetDevice( c->device );
            (void)hipStreamSynchronize( c->stream );
            (void)hipFree( c->fbOwn );
            c->fbOwn = nullptr;
            c->fbOwnPixels = 0;
        }
    }
    return VRC_OK;
}

int vrc_get_framebuffer( vrc_ctx* c, void** deviceRgba, uint32_t* width, uint32_t* height )
{
    if( !c )
        return fail( VRC_EINVAL, "vrc_get_framebuffer: ctx is NULL" );
    {
        const int rc = resolve_clear( c );
        if( rc != VRC_OK )
            return rc;
    }
    if( deviceRgba ) *deviceRgba = ctx_fb( c );
    if( width ) *width = c->fbW;
    if( height ) *height = c->fbH;
    return VRC_OK;
}

/* node table + brick grid: vrc_tables.h */
static int ensure_capacity( vrc_ctx* c, size_t nNodes, size_t nGrid )
{
    if( nNodes > c->dNodesCap )
    {
        VRC_HIP_CHECK( hipStreamSynchronize( c->stream ) );
        if( c->dNodes ) VRC_HIP_CHECK( hipFree( c->dNodes ) );
        c->dNodes = nullptr;
        c->dNodesCap = 0;
        const size_t cap = std::max< size_t >( nNodes, 1024 ); /* sized to n (fixes Q8) */
        VRC_HIP_CHECK( hipMalloc( &c->dNodes, cap * sizeof( vrc_dev_node ) ) );
        c->dNodesCap = cap;
    }
    if( nGrid > c->dGridCap )
    {
        VRC_HIP_CHECK( hipStreamSynchronize( c->stream ) );
        if( c->dGrid ) VRC_HIP_CHECK( hipFree( c->dGrid ) );
        c->dGrid = nullptr;
        c->dGridCap = 0;
        const size_t cap = std::max< size_t >( nGrid, 4096 );
        VRC_HIP_CHECK( hipMalloc( &c->dGrid, cap * sizeof( int32_t ) ) );
        c->dGridCap = cap;
    }
    const size_t stageBytes = nNodes * sizeof( vrc_dev_node ) + nGrid * sizeof( int32_t );
    if( stageBytes > c->hStageCap )
    {
        VRC_HIP_CHECK( hipStreamSynchronize( c->stream ) );
        if( c->hStage ) VRC_HIP_CHECK( hipHostFree( c->hStage ) );
        c->hStage = nullptr;
        c->hStageCap = 0;
        const size_t cap = ( std::max< size_t >( stageBytes, 1 << 18 ) + 255 ) / 256 * 256;
        VRC_HIP_CHECK( hipHostMalloc( &c->hStage, cap * kNodeStages ) );
        c->hStageCap = cap;
        for( int i = 0; i < kNodeStages; ++i )
            c->stageUsed[i] = false;
    }
    return VRC_OK;
}

int vrc_render( vrc_ctx* c, const vrc_view_data* view, const vrc_node_data* nodes, uint32_t nNodes,
                const vrc_render_data* render, vrc_pool* pool )
{
    if( !c || !view || !render || !pool )
        return fail( VRC_EINVAL, "vrc_render: NULL argument" );
    if( nNodes > 0 && !nodes )
        return fail( VRC_EINVAL, "vrc_render: nodes is NULL" );
    if( pool->device != c->device )
        return fail( VRC_EINVAL, "vrc_render: pool lives on another device" );
    if( !ctx_fb( c ) || c->fbW == 0 )
        return fail( VRC_EINVAL, "vrc_render: vrc_pre_render has not been called" );
    if( view->glViewport[2] != c->fbW ||
        ( c->rowMap.empty() ? view->glViewport[3] : (uint32_t)c->rowMap.size() ) != c->fbH )
        return fail( VRC_EINVAL, "vrc_render: viewport differs from the pixel buffer" );
    if( render->samplesPerRay == 0 )
        return fail( VRC_EINVAL, "vrc_render: samplesPerRay is 0" );
    if( render->dataSourceRange[1] == render->dataSourceRange[0] )
        return fail( VRC_EINVAL, "vrc_render: empty data source range" );
    if( nNodes == 0 ) /* CudaRaycastRenderer.cpp:157-158 */
        return VRC_OK;
    VRC_HIP_CHECK( hipSetDevice( c->device ) );

    /* classified table, rebuilt only when one of its inputs changed */
    vrc_lut_params lp;
    lp.rangeMin = render->dataSourceRange[0];
    lp.rangeMax = render->dataSourceRange[1];
    lp.alphaCorrection = (float)render->maxSamplesPerRay / (float)render->samplesPerRay;
    lp.fracBits = (int)c->optTfFracBits;
    const bool linear = c->optFilter == VRC_FILTER_TRILINEAR;
    /* samples classified one by one (padded transfer function in the table buffer) whenever the
     * 257-entry classified table cannot be used: continuous or 16-bit densities */
    const bool classify = linear || pool->elemBytes != 1;
    /* per-ray LOD: one classified table per level, opacity exponent doubled per level (a level-j
     * brick is sampled with step * 2^j); always all levels, so the table does not depend on the list */
    const uint32_t lutLevels = ( c->rayLod && !classify ) ? (uint32_t)VRC_MAX_LOD_LEVELS : 1u;
    if( !c->lutValid || c->lutTfVersion != c->tfVersion || c->lutLinear != classify ||
        c->lutLevels != lutLevels || std::memcmp( &lp, &c->lutParams, sizeof( lp ) ) != 0 )
    {
        for( uint32_t j = 0; j < lutLevels; ++j )
        {
            vrc_lut_params lj = lp;
            lj.alphaCorrection = lp.alphaCorrection * (float)( 1u << j );
            VRC_HIP_CHECK( vrc_launch_build_lut( c->dTf, c->dLut + j * VRC_LUT_ENTRIES, lj, classify,
                                                 c->stream ) );
        }
        c->lutParams = lp;
        c->lutLinear = classify;
        c->lutLevels = lutLevels;
        c->lutTfVersion = c->tfVersion;
        c->lutValid = true;
        /* the largest classified opacity of a sample (same function as the device table, evaluated on the
         * pinned copy of the transfer function): decides whether early ray termination can occur at all */
        c->lutMaxAlpha = 1.0f;
        if( !classify )
        {
            float m = 0.0f;
            for( uint32_t d = 0; d < 256u; ++d )
                m = std::max( m, vrc_lut_entry( c->hTf, d, lp ).w );
            c->lutMaxAlpha = m;
        }
    }

    /* node table + grid, re-derived and re-uploaded only when the node list changed
     * (the reference re-uploads synchronously every pass, Renderer.cu:259-267) */
    const bool sameNodes = c->cachedPoolUid == pool->uid && c->cachedNodes.size() == nNodes &&
                           c->cachedRayLod == c->rayLod &&
                           std::memcmp( c->cachedNodes.data(), nodes,
                                        nNodes * sizeof( vrc_node_data ) ) == 0;
    if( !sameNodes )
    {
        vrc_host_tables t;
        vrc_atlas_geom geom;
        for( int a = 0; a < 3; ++a )
        {
            geom.atlasDim[a] = pool->atlasDim[a];
            geom.slotDim[a] = pool->slotDim[a];
        }
        for( int a = 0; a < 3; ++a )
            geom.slots[a] = pool->slots[a];
        vrc_build_tables( geom, nodes, nNodes, t );
        if( c->rayLod )
            vrc_build_lod_tables( geom, nodes, nNodes, t );
        const int rc = ensure_capacity( c, t.nodes.size(), t.grid.size() );
        if( rc != VRC_OK )
            return rc;
        const uint32_t stage = c->stageNext++ % kNodeStages;
        if( c->stageUsed[stage] ) /* the copies issued from this area four node lists ago */
            VRC_HIP_CHECK( hipEventSynchronize( c->stageEvent[stage] ) );
        if( !c->stageEvent[stage] )
            VRC_HIP_CHECK( hipEventCreateWithFlags( &c->stageEvent[stage], hipEventDisableTiming ) );
        uint8_t* h = (uint8_t*)c->hStage + (size_t)stage * c->hStageCap;
        const size_t nb = t.nodes.size() * sizeof( vrc_dev_node );
        const size_t gb = t.grid.size() * sizeof( int32_t );
        std::memcpy( h, t.nodes.data(), nb );
        VRC_HIP_CHECK( hipMemcpyAsync( c->dNodes, h, nb, hipMemcpyHostToDevice, c->stream ) );
        if( gb )
        {
            std::memcpy( h + nb, t.grid.data(), gb );
            VRC_HIP_CHECK(
                hipMemcpyAsync( c->dGrid, h + nb, gb, hipMemcpyHostToDevice, c->stream ) );
        }
        VRC_HIP_CHECK( hipEventRecord( c->stageEvent[stage], c->stream ) );
        c->stageUsed[stage] = true;
        c->cachedNodes.assign( nodes, nodes + nNodes );
        c->cachedPoolUid = pool->uid;
        c->cachedGridOk = t.gridOk && !c->rayLod; /* the grid buffer holds the per-level tables */
        c->cachedOneCell = t.oneCellPerBrick;
        c->cachedRayLod = c->rayLod;
        c->cachedLodOk = t.lodOk;
        c->cachedLodLevels = t.lodLevels;
        c->cachedFinestVoxel = t.finestVoxelWorld;
        c->cachedClamp = t.clamp;
        c->cachedGridFrame = t.g;
    }

    if( c->rayLod )
    {
        if( !c->cachedLodOk )
            return fail( VRC_EHIERARCHY, "vrc_render: per-ray LOD needs a brick hierarchy (every level a regular grid of bricks, one brick per cell)" );
        if( c->optVariant != VRC_VARIANT_CUDARAYCASTER )
            return fail( VRC_EINVAL, "vrc_render: per-ray LOD is defined for the cudaRaycaster variant only" );
        /* AUTO: the LDS-staged form for the trilinear filter where it applies, else the gather form; GRID_DDA asks for
         * the gather form, LDS for the staged one */
        if( c->optKernel == VRC_KERNEL_REFERENCE_ORDER )
            return fail( VRC_EINVAL, "vrc_render: per-ray LOD walks the hierarchy; VRC_OPT_KERNEL = AUTO, GRID_DDA (gathers), LDS or PACKED" );
    }
    bool useDda = c->cachedGridOk;
    /* AUTO keeps the reference's frame: bricks of one size are met by the grid walk in the reference's
     * order; a list that mixes brick sizes (an LOD cut) is composited in the host's centre-distance order,
     * which is not a visibility order for every ray (quirk Q6) -- the literal loop reproduces that as long
     * as testing every brick per ray is affordable.  GRID_DDA can always be asked for (true visibility
     * order); the trilinear filter (extension, no reference frame to keep) stays on the grid. */
    if( c->optKernel == VRC_KERNEL_AUTO && useDda && !c->cachedOneCell && !linear &&
        nNodes <= VRC_REFERENCE_ORDER_MAX_NODES )
        useDda = false;
    /* glRaycaster with more than one sample per pixel (fragRaycast.glsl:121-129): the pixel is averaged brick by
     * brick in the host's order -- the reference-order loop is the only form that has a "brick by brick" */
    const bool glSuper = c->optVariant == VRC_VARIANT_GLRAYCASTER && render->samplesPerPixel > 1u;
    if( glSuper )
    {
        if( c->optKernel == VRC_KERNEL_GRID_DDA || c->optKernel == VRC_KERNEL_LDS )
            return fail( VRC_EINVAL, "vrc_render: the glRaycaster variant with samplesPerPixel > 1 is rendered by the "
                                     "reference-order kernel (VRC_OPT_KERNEL = AUTO or REFERENCE_ORDER)" );
        useDda = false;
    }
    if( c->optKernel == VRC_KERNEL_REFERENCE_ORDER )
        useDda = false;
    else if( c->optKernel == VRC_KERNEL_GRID_DDA && !c->cachedGridOk && !c->rayLod )
        return fail( VRC_EINVAL, "vrc_render: node set is not grid-aligned; GRID_DDA unavailable" );
    /* LDS-staged kernel: brick-grid DDA + unclamped sampler (overlap >= 1).  AUTO takes it for
     * the trilinear filter (eight taps per sample), the gather kernel for point sampling. */
    /* its trilinear form takes the transfer-function texel and CUDA's 1.8 fixed-point lerp weight out of one
     * float -> integer conversion (vrc_kernels_lds.hip: lds_classify): other weight widths use the gather form */
    /* slot-local positions in 8.24 fixed point (fixed-point stepping, the per-axis address tables, the LDS kernel's boxes)
     * need every slot dimension to fit eight bits; pool creation bounds a slot's VOLUME only (2^24 voxels), so a flat
     * or long brick can exceed it: such pools march with float positions */
    const bool slotsFit8Bits = pool->slotDim[0] <= 248u && pool->slotDim[1] <= 248u && pool->slotDim[2] <= 248u;
    /* 16-bit voxels: its trilinear form only (a 16-bit density does not index the classified table of the point form) */
    const bool ldsVoxels = slotsFit8Bits && ( pool->elemBytes == 1 || ( pool->elemBytes == 2 && linear ) );
    /* atlases of more than 2^32 voxels (64-bit slot bases): its trilinear form only */
    const bool ldsEligible = c->cachedGridOk && !c->cachedClamp && ldsVoxels && ( !pool->bigAtlas || linear ) &&
                             ( !linear || c->optTfFracBits == 8 );
    if( !c->rayLod && c->optKernel == VRC_KERNEL_LDS && !ldsEligible )
        return fail( VRC_EINVAL, "vrc_render: the LDS kernel needs a grid-aligned node set of 8-bit bricks in an atlas of at most 2^32 voxels (trilinear: 8- or 16-bit, any atlas, VRC_OPT_TF_FRAC_BITS = 8) with overlap >= 1" );
    /* per-ray LOD: the staged kernel's trilinear form only (samples classified one by one: no table per level) */
    const bool ldsLodEligible = c->rayLod && linear && !c->cachedClamp && ldsVoxels && !pool->bigAtlas &&
                                c->optTfFracBits == 8;
    if( c->rayLod && c->optKernel == VRC_KERNEL_LDS && !ldsLodEligible )
        return fail( VRC_EINVAL, "vrc_render: under per-ray LOD the LDS kernel needs the trilinear filter, 8- or 16-bit bricks with overlap >= 1 and VRC_OPT_TF_FRAC_BITS = 8" );
    /* tap-packed atlas (VRC_KERNEL_PACKED; vrc_core.h): the trilinear filter as two gathers per sample.  Needs what its
     * positions and its classifier need -- 8- or 16-bit bricks with overlap >= 1 in slots of at most 248 voxels a side,
     * VRC_OPT_TF_FRAC_BITS = 8, fixed-point stepping -- and 2.25 times the atlas in device memory; either brick
     * enumeration (grid walk where the node set is grid-aligned, else the reference-order loop) */
    const bool packedEligible = linear && !glSuper && !c->cachedClamp && slotsFit8Bits &&
                                pool_packed_possible( pool ) && c->optTfFracBits == 8 && c->optStepping != 0;
    bool usePacked = false;
    if( c->optKernel == VRC_KERNEL_PACKED )
    {
        if( !packedEligible )
            return fail( VRC_EINVAL, "vrc_render: the packed kernel needs the trilinear filter on 8- or 16-bit bricks with overlap >= 1 (slots of at most 248 voxels a side), VRC_OPT_TF_FRAC_BITS = 8, fixed-point stepping" );
        if( !pool_enable_packed( pool ) )
            return fail( VRC_ENOMEM, "vrc_render: no device memory for the tap-packed atlas (2.25 times the brick atlas)" );
        usePacked = true;
    }
    else if( c->optKernel == VRC_KERNEL_AUTO && packedEligible && c->optPackedAtlas )
        /* measured on C2: 1.2 against 1.7 ms along the axis, 1.5 against 2.4 at 30/20 degrees;
         * under per-ray LOD 1.4-2.1 x the staged form (profiles/r4_c5_trilinear_three_forms.txt); without the memory for
         * it the frame takes the staged form below */
        usePacked = pool_enable_packed( pool );
    const bool useLds = c->rayLod ? ( ldsLodEligible && c->optKernel != VRC_KERNEL_GRID_DDA && !usePacked )
                                  : !glSuper && !usePacked && ( c->optKernel == VRC_KERNEL_LDS ||
                                                  ( c->optKernel == VRC_KERNEL_AUTO && linear && ldsEligible ) );

    vrc_raycast_args a;
    std::memset( &a, 0, sizeof( a ) );
    vrc_frame& f = a.frame;
    {
        vrc_atlas_geom geom;
        for( int i = 0; i < 3; ++i )
        {
            geom.atlasDim[i] = pool->atlasDim[i];
            geom.slotDim[i] = pool->slotDim[i];
        }
        for( int a = 0; a < 3; ++a )
            geom.slots[a] = pool->slots[a];
        /* the GLSL twin casts through the pixel centre (gl_FragCoord, fragRaycast.glsl:127), the
         * CUDA kernel through the pixel corner (Renderer.cu:106-112) */
        const float centre = c->optVariant == VRC_VARIANT_GLRAYCASTER ? 0.5f : 0.0f;
        vrc_fill_frame( f, *view, *render, geom, c->cachedGridFrame, c->planes, c->nPlanes, nNodes,
                        c->fbW, c->fbH, centre, centre );
        f.rowMap = c->rowMap.empty() ? nullptr : c->dRowMap;
        f.variant = c->optVariant == VRC_VARIANT_GLRAYCASTER ? VRC_VARIANT_GL : VRC_VARIANT_CUDA;
        f.samplesPerPixel = ( f.variant == VRC_VARIANT_GL && render->samplesPerPixel > 1u ) ? render->samplesPerPixel : 1u;
        if( c->rayLod )
        {
            f.lodLevels = c->cachedLodLevels;
            f.lodBase = (float)( c->cachedFinestVoxel /
                                 ( (double)c->rayLodSse * (double)c->rayLodWorldPerPixel ) );
        }
    }

    /* tile schedule, recomputed only when the frame constants changed */
    a.tileOrder = nullptr;
    if( c->optTileOrder )
    {
        /* one entry per schedule slot: four per 2x2 block of tiles (vrc_internal.h) */
        const size_t nTiles = vrc_schedule_slots( ( c->fbW + VRC_TILE_W - 1 ) / VRC_TILE_W,
                                                  ( c->fbH + VRC_TILE_H - 1 ) / VRC_TILE_H );
        if( nTiles > c->dTileOrderCap )
        {
            VRC_HIP_CHECK( hipStreamSynchronize( c->stream ) );
            if( c->dTileOrder ) VRC_HIP_CHECK( hipFree( c->dTileOrder ) );
            c->dTileOrder = nullptr;
            c->dTileOrderCap = 0;
            /* order[nTiles] | scratch[VRC_TILE_SCRATCH_WORDS] | bucket[nTiles] (bytes) */
            VRC_HIP_CHECK( hipMalloc( &c->dTileOrder, ( nTiles + VRC_TILE_SCRATCH_WORDS ) * sizeof( uint32_t ) + nTiles ) );
            c->dTileOrderCap = nTiles;
            c->tileOrderValid = false;
        }
        /* The schedule is a heuristic (any permutation of the tiles renders the same frame): a camera
         * that moved a little keeps the last one -- same pixel buffer and volume box, eye and ray
         * matrices within 2 % -- for at most 15 frames in a row; two small kernels and a memset
         * less per frame of an orbit. */
        bool reuse = c->tileOrderValid;
        if( reuse && std::memcmp( &c->tileOrderFrame, &f, sizeof( f ) ) != 0 )
        {
            const vrc_frame& o = c->tileOrderFrame;
            reuse = c->tileOrderReused < 15u && o.width == f.width && o.height == f.height && o.vpW == f.vpW &&
                    o.vpH == f.vpH && o.vpX == f.vpX && o.vpY == f.vpY && o.pixelOffX == f.pixelOffX &&
                    o.pixelOffY == f.pixelOffY && o.rowMap == f.rowMap && o.nPlanes == f.nPlanes &&
                    std::memcmp( o.planes, f.planes, sizeof( f.planes ) ) == 0 &&
                    std::memcmp( o.aabbMin, f.aabbMin, sizeof( f.aabbMin ) ) == 0 &&
                    std::memcmp( o.aabbMax, f.aabbMax, sizeof( f.aabbMax ) ) == 0 &&
                    std::memcmp( o.invProj, f.invProj, sizeof( f.invProj ) ) == 0;
            for( int i = 0; reuse && i < 3; ++i )
                reuse = std::fabs( o.eye[i] - f.eye[i] ) <= 0.02f;
            for( int i = 0; reuse && i < 16; ++i )
                reuse = std::fabs( o.invView[i] - f.invView[i] ) <= 0.02f;
            if( reuse )
                ++c->tileOrderReused;
        }
        if( !reuse )
        {
            c->tileOrderReused = 0;
            VRC_HIP_CHECK( vrc_launch_tile_order( f, c->dTileOrder, c->dTileOrder + c->dTileOrderCap,
                                                  (uint8_t*)( c->dTileOrder + c->dTileOrderCap + VRC_TILE_SCRATCH_WORDS ),
                                                  c->stream ) );
            std::memcpy( &c->tileOrderFrame, &f, sizeof( f ) );
            c->tileOrderValid = true;
        }
        a.tileOrder = c->dTileOrder;
    }

    /* fold the frame's clear into this march (after the tile-schedule cache compared frames) */
    f.clearFirst = c->clearPending ? 1u : 0u;
    c->clearPending = false;

    a.nodes = c->dNodes;
    a.gridTable = ( useDda || c->rayLod ) ? c->dGrid : nullptr;
    a.atlas = usePacked ? pool->dPacked : pool->dAtlas;
    a.packed = usePacked;
    /* a packed atlas of more than 4 GiB, or of a pool of more than 2^32 voxels: 64-bit lane pointers (the BIG instances of
     * the packed modes) */
    a.packedWide = usePacked && ( pool->bigAtlas || pool_packed_bytes( pool ) > 0xFFFFFFFFull );
    a.lut = c->dLut;
    a.pixelBuffer = ctx_fb( c );
    a.sampleCounter = c->optCount ? c->dCounter : nullptr;
    a.clamp = c->cachedClamp;
    a.gridDda = useDda;
    a.fixedStepping = c->optStepping != 0 && slotsFit8Bits;
    a.linear = linear;
    a.elemBytes = pool->elemBytes;
    a.bigAtlas = pool->bigAtlas;
    /* depth split: only where it is exact.  (i) The far half cannot know the near half's opacity, so early ray
     * termination must be impossible: (1 - largest classified alpha)^(most samples a ray can take) stays above
     * 1 - 0.999 (a ray crosses at most sqrt(3) world units -- the volume's longest edge is 1 -- plus one
     * restart sample per brick).  (ii) The frame starts from zero (first pass: the clear is folded in).
     * (iii) the table-driven point-sampling walk kernel. */
    a.depthSplit = false;
    if( c->optDepthSplit && useDda && !useLds && !c->rayLod && !linear && pool->elemBytes == 1 && !pool->bigAtlas &&
        !c->cachedClamp && c->optStepping != 0 && slotsFit8Bits && f.clearFirst )
    {
        const double nMax = 1.7320508 * (double)render->samplesPerRay +
                            3.0 * ( f.gridDim[0] + f.gridDim[1] + f.gridDim[2] ) + 8.0;
        /* with a margin: the kernel accumulates the opacity in float with contracted multiply-adds and regroups the
         * far half's additions, ~1e-6 absolute after 2000 samples = 1e-3 of the 0.001 of transmittance left at
         * the threshold; the bound must clear the threshold by ten times that (0.01 in the logarithm), so that a
         * frame at the limit takes the single-wave kernel rather than a split that might meet an early exit */
        a.depthSplit = VRC_TILE_W == 8u && c->lutMaxAlpha < 1.0f &&
                       nMax * std::log1p( -(double)c->lutMaxAlpha ) > std::log( 1.0 - 0.999 ) + 0.01;
    }
    /* ray compaction: the table-driven point-sampling walk kernel; packed 16-bit pixel coordinates */
    a.ertParts = 0;
    a.rayList = nullptr;
    /* (the same predicate as the launcher's, vrc_launch_raycast: what vrc_get_ray_counts reports is what ran) */
    if( c->optErtParts > 1 && !a.depthSplit && useDda && !useLds && !c->rayLod && !linear && pool->elemBytes == 1 &&
        !pool->bigAtlas && !c->cachedClamp && c->optStepping != 0 && slotsFit8Bits && c->fbW < 65536u && c->fbH < 65536u &&
        VRC_TILE_W == 8u )
    {
        const size_t pixels = (size_t)c->fbW * c->fbH;
        if( pixels > c->dRayListCap )
        {
            VRC_HIP_CHECK( hipStreamSynchronize( c->stream ) );
            if( c->dRayList ) VRC_HIP_CHECK( hipFree( c->dRayList ) );
            c->dRayList = nullptr;
            c->dRayListCap = 0;
            VRC_HIP_CHECK( hipMalloc( &c->dRayList, ( VRC_MAX_ERT_PARTS + 2 * pixels ) * sizeof( uint32_t ) ) );
            c->dRayListCap = pixels;
        }
        a.ertParts = (int)c->optErtParts;
        a.rayList = c->dRayList;
    }
    c->lastErtParts = a.ertParts;
    /* grey transfer function, frame starting from zero: two-float table entries, the same bits (VRC_MODE_GREY) */
    a.greyTable = c->optGreyTable && c->tfGrey && f.clearFirst && !glSuper; /* (a sub-ray starts from the pixel so far) */
    a.classifier = vrc_make_classifier( lp );

    /* order the march after every brick upload issued so far (fixes quirk Q9) */
    {
        std::lock_guard< std::mutex > lock( pool->mutex );
        if( pool->hasUpload )
            VRC_HIP_CHECK( hipStreamWaitEvent( c->stream, pool->lastUpload, 0 ) );
    }
    if( c->optCount )
        VRC_HIP_CHECK( hipMemsetAsync( c->dCounter, 0, sizeof( unsigned long long ), c->stream ) );
    if( c->optTiming && c->evUsed == c->evPairs.size() )
    {
        if( c->evPairs.size() >= 4096 )
        {
            /* nobody has read the timings for 4096 launches: fold them into running sums (waits for the
             * launches still in flight, once per 4096) so that vrc_get_stats still accounts for every one */
            VRC_HIP_CHECK( hipEventSynchronize( c->evPairs[c->evUsed - 1].second ) );
            for( size_t i = 0; i < c->evUsed; ++i )
            {
                float ms = 0.f;
                VRC_HIP_CHECK( hipEventElapsedTime( &ms, c->evPairs[i].first, c->evPairs[i].second ) );
                c->evFoldedMs += ms;
            }
            c->evFoldedLaunches += (uint32_t)c->evUsed;
            c->evUsed = 0;
        }
        else
        {
            hipEvent_t a0 = nullptr, a1 = nullptr;
            VRC_HIP_CHECK( hipEventCreate( &a0 ) );
            VRC_HIP_CHECK( hipEventCreate( &a1 ) );
            c->evPairs.push_back( { a0, a1 } );
        }
    }
    const std::pair< hipEvent_t, hipEvent_t > evp =
        c->optTiming ? c->evPairs[c->evUsed++] : std::pair< hipEvent_t, hipEvent_t >( nullptr, nullptr );
    vrc_internal_note_kernel_fn( nullptr, 0, 0 ); /* set again by the launchers that report their occupancy */
    if( c->optTiming )
        VRC_HIP_CHECK( hipEventRecord( evp.first, c->stream ) );
    VRC_HIP_CHECK( useLds      ? vrc_launch_raycast_lds( a, c->stream ) /* (also its per-ray LOD form) */
                   : c->rayLod ? vrc_launch_raycast_raylod( a, c->stream )
                               : vrc_launch_raycast( a, c->stream ) );
    if( c->optTiming )
        VRC_HIP_CHECK( hipEventRecord( evp.second, c->stream ) );
    {
        /* render fence of this context on the pool (see pool_upload) */
        std::lock_guard< std::mutex > lock( pool->mutex );
        hipEvent_t fence = nullptr;
        for( const auto& f : pool->renderFences )
            if( f.ctx == c )
                fence = f.event;
        if( !fence )
        {
            VRC_HIP_CHECK( hipEventCreateWithFlags( &fence, hipEventDisableTiming ) );
            pool->renderFences.push_back( { c, fence } );
        }
        VRC_HIP_CHECK( hipEventRecord( fence, c->stream ) );
    }
    if( c->optCount )
        VRC_HIP_CHECK( hipMemcpyAsync( c->hCounter, c->dCounter, sizeof( unsigned long long ),
                                       hipMemcpyDeviceToHost, c->stream ) );
    c->timed = true; /* a render has happened: vrc_get_stats has something to report */
    c->stats.kernel_variant =
        c->rayLod ? VRC_KERNEL_RAY_LOD
        : useLds  ? VRC_KERNEL_LDS
        : usePacked ? VRC_KERNEL_PACKED
                  : ( useDda ? VRC_KERNEL_GRID_DDA : VRC_KERNEL_REFERENCE_ORDER );
    for( int i = 0; i < 3; ++i )
        c->stats.grid_dims[i] = ( useDda || c->rayLod ) ? (uint32_t)f.gridDim[i] : 0u;
    return VRC_OK;
}

int vrc_post_render( vrc_ctx* c, float* hostRgba )
{
    if( !c )
        return fail( VRC_EINVAL, "vrc_post_render: ctx is NULL" );
    VRC_HIP_CHECK( hipSetDevice( c->device ) );
    {
        const int rc = resolve_clear( c ); /* a frame without a march still ends cleared */
        if( rc != VRC_OK )
            return rc;
    }
    if( hostRgba )
    {
        if( !ctx_fb( c ) || c->fbW == 0 )
            return fail( VRC_EINVAL, "vrc_post_render: no pixel buffer" );
        VRC_HIP_CHECK( hipMemcpyAsync( hostRgba, ctx_fb( c ),
                                       (size_t)c->fbW * c->fbH * sizeof( vrc_f4 ),
                                       hipMemcpyDeviceToHost, c->stream ) );
        VRC_HIP_CHECK( hipStreamSynchronize( c->stream ) );
    }
    return VRC_OK;
}

int vrc_synchronize( vrc_ctx* c )
{
    if( !c )
        return fail( VRC_EINVAL, "vrc_synchronize: ctx is NULL" );
    VRC_HIP_CHECK( hipSetDevice( c->device ) );
    {
        const int rc = resolve_clear( c );
        if( rc != VRC_OK )
            return rc;
    }
    VRC_HIP_CHECK( hipStreamSynchronize( c->stream ) );
    return VRC_OK;
}

int vrc_get_ray_counts( vrc_ctx* c, uint32_t counts[8], int* parts )
{
    if( !c || !counts || !parts )
        return fail( VRC_EINVAL, "vrc_get_ray_counts: NULL argument" );
    static_assert( VRC_MAX_ERT_PARTS == 8, "vrc_get_ray_counts reports 8 entries" );
    VRC_HIP_CHECK( hipSetDevice( c->device ) );
    for( int i = 0; i < 8; ++i )
        counts[i] = 0;
    *parts = c->lastErtParts;
    if( c->lastErtParts > 1 )
    {
        VRC_HIP_CHECK( hipStreamSynchronize( c->stream ) );
        VRC_HIP_CHECK( hipMemcpy( counts, c->dRayList, VRC_MAX_ERT_PARTS * sizeof( uint32_t ), hipMemcpyDeviceToHost ) );
    }
    return VRC_OK;
}

int vrc_get_stats( vrc_ctx* c, vrc_stats* out )
{
    if( !c || !out )
        return fail( VRC_EINVAL, "vrc_get_stats: NULL argument" );
    VRC_HIP_CHECK( hipSetDevice( c->device ) );
    c->stats.kernel_ms_sum = 0.0;
    c->stats.kernel_launches = 0;
    if( c->evUsed > 0 )
    {
        VRC_HIP_CHECK( hipEventSynchronize( c->evPairs[c->evUsed - 1].second ) );
        float ms = 0.f;
        for( size_t i = 0; i < c->evUsed; ++i )
        {
            VRC_HIP_CHECK( hipEventElapsedTime( &ms, c->evPairs[i].first, c->evPairs[i].second ) );
            c->stats.kernel_ms_sum += ms;
        }
        c->stats.kernel_launches = (uint32_t)c->evUsed;
        c->stats.kernel_ms = ms; /* the last launch */
        c->evUsed = 0;
        c->stats.kernel_ms_sum += c->evFoldedMs;
        c->stats.kernel_launches += c->evFoldedLaunches;
        c->evFoldedMs = 0.0;
        c->evFoldedLaunches = 0;
    }
    else if( !c->optTiming )
        c->stats.kernel_ms = 0.f;
    if( c->timed )
    {
        /* the sample counter of the last vrc_render (VRC_OPT_COUNT_SAMPLES) */
        if( c->optCount )
        {
            VRC_HIP_CHECK( hipStreamSynchronize( c->stream ) );
            c->stats.samples = *c->hCounter;
        }
        else
            c->stats.samples = 0;
    }
    *out = c->stats;
    return VRC_OK;
}

} /* extern "C" */

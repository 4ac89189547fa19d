/* hip_plugin.cpp -- HipTexturePool, HipTextureObject, HipRaycastRenderer, HipRaycastPipeline:
 * the host plugin of the MI355X raycaster.  Follows the renderers/cudaRaycaster sources of the
 * reference function by function (citations inline); device work is the C ABI of vrc_hip.h. */
#include "livre_hip/hip.h"

#include <unordered_set>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <map>
#include <queue>
#include <thread>

#include "vrc_hip.h"

extern "C" int LunchboxPluginGetVersion() { return 1; } /* LIVRECORE_VERSION_ABI, CudaRaycastPipeline.cpp:50-51 */
extern "C" bool LunchboxPluginRegister() { return true; } /* CudaRaycastPipeline.cpp:53-54 */

namespace livre
{
namespace
{
const Vector3f INVALID_SLOT_POSITION( -1.0f ); /* CudaTextureObject.cpp:36-39 */
const uint32_t maxSamplesPerRay = 32;  /* CudaRaycastRenderer.cpp:65 (the opacity-correction reference) */
const uint32_t minSamplesPerRay = 512; /* CudaRaycastRenderer.cpp:66 */
const uint32_t SH_UINT = 0u, SH_INT = 1u, SH_FLOAT = 2u;
const size_t nAsyncUploadThreads = 1; /* CudaRaycastPipeline.cpp:60-63 */

/** data loaders feeding the texture uploader: 2 in the reference (CudaRaycastPipeline.cpp:60-63);
 * here half the host cores, between 2 and 8, or LIVRE_HIP_UPLOAD_THREADS (one loader moves
 * ~7 GB/s of mem:// bricks, the upload path behind them takes ~35 GB/s) */
size_t configuredUploadThreads()
{
    const char* v = std::getenv( "LIVRE_HIP_UPLOAD_THREADS" );
    if( v && *v )
    {
        const long n = std::strtol( v, nullptr, 10 );
        if( n >= 1 && n <= 64 )
            return size_t( n );
    }
    const size_t hw = std::thread::hardware_concurrency();
    return std::min< size_t >( 8, std::max< size_t >( 2, hw / 2 ) );
}

PluginRegisterer< HipRaycastRenderer, const std::string& > rendererRegisterer;
PluginRegisterer< HipRaycastPipeline, const std::string& > pipelineRegisterer;

std::atomic< int > g_device( 0 );
std::mutex g_deviceMutex;
/* the context the texture pools of a device are created on: one per device (an application on a second
 * device must not get its atlas on the first), shared by the pools of that device and destroyed with the
 * last of them (not at static-destruction time: the HIP runtime may be gone by then) */
struct DeviceContext
{
    vrc_ctx* ctx = nullptr;
    size_t users = 0;
};
std::map< int, DeviceContext > g_deviceContexts;

/* the device layer this plugin was compiled against (a developer build of it reports the negated version) */
void checkDeviceLayerAbi()
{
    const int v = vrc_abi_version();
    if( v != VRC_ABI_VERSION && v != -VRC_ABI_VERSION )
        throw std::runtime_error( "libvrc_hip.so has ABI version " + std::to_string( v ) + ", this plugin needs " +
                                  std::to_string( VRC_ABI_VERSION ) );
}

vrc_ctx* acquireDeviceContext( int* deviceOut )
{
    checkDeviceLayerAbi();
    std::lock_guard< std::mutex > lock( g_deviceMutex );
    const int device = g_device.load();
    DeviceContext& d = g_deviceContexts[device];
    if( !d.ctx )
        throwOnVrcError( vrc_ctx_create( device, &d.ctx ), "vrc_ctx_create" );
    ++d.users;
    *deviceOut = device;
    return d.ctx;
}

void releaseDeviceContext( int device )
{
    std::lock_guard< std::mutex > lock( g_deviceMutex );
    const auto it = g_deviceContexts.find( device );
    if( it == g_deviceContexts.end() || it->second.users == 0 )
        return;
    if( --it->second.users == 0 )
    {
        vrc_ctx_destroy( it->second.ctx );
        g_deviceContexts.erase( it );
    }
}

/** LIVRE_HIP_PROFILE=1: host time of the stages of a frame, summed and printed when the pipeline goes */
struct StageClock
{
    enum Stage { VisibleSet, Upload, PreRender, SortAndFill, DeviceCalls, PostRender, nStages };
    static bool enabled()
    {
        static const bool on = std::getenv( "LIVRE_HIP_PROFILE" ) != nullptr;
        return on;
    }
    static double& sum( Stage s )
    {
        static double sums[nStages] = { 0, 0, 0, 0, 0, 0 };
        return sums[s];
    }
    static size_t& frames()
    {
        static size_t n = 0;
        return n;
    }
    static size_t& orderKept()
    {
        static size_t n = 0;
        return n;
    }
    static size_t& orderSorted()
    {
        static size_t n = 0;
        return n;
    }
    explicit StageClock( Stage s ) : _stage( s ), _on( enabled() )
    {
        if( _on )
            _t0 = std::chrono::steady_clock::now();
    }
    ~StageClock()
    {
        if( _on )
            sum( _stage ) += std::chrono::duration< double, std::micro >( std::chrono::steady_clock::now() - _t0 ).count();
    }
    static void report()
    {
        if( !enabled() || frames() == 0 )
            return;
        static const char* names[nStages] = { "visible set", "upload / cache look-ups", "preRender", "sort + node table",
                                              "device calls", "postRender" };
        std::fprintf( stderr, "[livre_hip] host time per frame over %zu frames (render thread; repeated frames skip the first two):", frames() );
        for( int i = 0; i < nStages; ++i )
            std::fprintf( stderr, " %s %.1f us;", names[i], sum( Stage( i ) ) / double( frames() ) );
        std::fprintf( stderr, " new views of the same bricks: order kept %zu times, sorted anew %zu times\n", orderKept(),
                      orderSorted() );
    }
    Stage _stage;
    bool _on;
    std::chrono::steady_clock::time_point _t0;
};

/** std::thread stand-in for tuyau::PushExecutor: n workers draining a task queue */
class Executor
{
public:
    explicit Executor( size_t nThreads ) : _stop( false ), _busy( 0 )
    {
        for( size_t i = 0; i < nThreads; ++i )
            _threads.emplace_back( [this] { run(); } );
    }
    ~Executor()
    {
        {
            std::lock_guard< std::mutex > lock( _mutex );
            _stop = true;
        }
        _cv.notify_all();
        for( std::thread& t : _threads )
            t.join();
    }
    void schedule( std::function< void() > task )
    {
        {
            std::lock_guard< std::mutex > lock( _mutex );
            _tasks.push( std::move( task ) );
        }
        _cv.notify_one();
    }
    void wait()
    {
        std::unique_lock< std::mutex > lock( _mutex );
        _idle.wait( lock, [this] { return _tasks.empty() && _busy == 0; } );
    }

private:
    void run()
    {
        for( ;; )
        {
            std::function< void() > task;
            {
                std::unique_lock< std::mutex > lock( _mutex );
                _cv.wait( lock, [this] { return _stop || !_tasks.empty(); } );
                if( _stop && _tasks.empty() )
                    return;
                task = std::move( _tasks.front() );
                _tasks.pop();
                ++_busy;
            }
            try
            {
                task();
            }
            catch( const std::exception& e )
            {
                /* tasks report their own failures to whoever waits for them; one that still leaks an
                 * exception must at least not vanish */
                std::fprintf( stderr, "[livre_hip] executor: task failed: %s\n", e.what() );
            }
            catch( ... )
            {
                std::fprintf( stderr, "[livre_hip] executor: task failed with a non-standard exception\n" );
            }
            {
                std::lock_guard< std::mutex > lock( _mutex );
                --_busy;
            }
            _idle.notify_all();
        }
    }
    std::vector< std::thread > _threads;
    std::queue< std::function< void() > > _tasks;
    std::mutex _mutex;
    std::condition_variable _cv, _idle;
    bool _stop;
    size_t _busy;
};

/** CudaRaycastRenderer.cpp:41-61 / CudaRaycastPipeline.cpp:107-127: |MV * box centre| */
float nodeDistance( const DataSource& dataSource, const Frustum& frustum, const NodeId& id )
{
    const LODNode lodNode = dataSource.getNode( id );
    return ( frustum.getMVMatrix() * lodNode.getWorldBox().getCenter() ).length();
}
}

void throwOnVrcError( int rc, const char* what )
{
    if( rc != VRC_OK )
        throw std::runtime_error( std::string( what ) + ": " + vrc_last_error() );
}

void setHipDevice( int device ) { g_device = device; }
int getHipDevice() { return g_device; }

/* ---- HipTexturePool: CudaTexturePool.cpp:31-122 --------------------------------------------- */
HipTexturePool::HipTexturePool( const DataSource& dataSource, size_t textureMemory )
    : _ctx( nullptr ), _pool( nullptr ), _device( 0 )
{
    _ctx = acquireDeviceContext( &_device );
    struct Release /* a constructor that throws runs no destructor */
    {
        int device;
        bool armed;
        ~Release()
        {
            if( armed )
                releaseDeviceContext( device );
        }
    } guard{ _device, true };
    const VolumeInformation& volInfo = dataSource.getVolumeInfo();
    bool isSigned = false, isFloat = false;
    switch( volInfo.dataType )
    {
    case DT_UINT8: case DT_UINT16: case DT_UINT32: break;
    case DT_INT8: case DT_INT16: case DT_INT32: isSigned = true; break;
    case DT_FLOAT: isFloat = true; break;
    case DT_UNDEFINED:
    default: throw std::runtime_error( "Undefined data type" );
    }
    const uint32_t maxBlock[3] = { volInfo.maximumBlockSize[0], volInfo.maximumBlockSize[1],
                                   volInfo.maximumBlockSize[2] };
    throwOnVrcError( vrc_pool_create( _ctx, volInfo.getBytesPerVoxel(), isSigned, isFloat,
                                      volInfo.compCount, maxBlock, textureMemory, &_pool ),
                     "vrc_pool_create" );
    guard.armed = false;
}

HipTexturePool::~HipTexturePool()
{
    vrc_pool_destroy( _pool );
    releaseDeviceContext( _device );
}

Vector3f HipTexturePool::copyToSlot( const unsigned char* ptr, const Vector3ui& size )
{
    const uint32_t s[3] = { size[0], size[1], size[2] };
    float slot[3];
    const int rc = vrc_pool_copy_to_slot( _pool, ptr, s, slot );
    if( rc == VRC_EFULL )
        return INVALID_SLOT_POSITION; /* TexturePool.cu:180-181 */
    throwOnVrcError( rc, "vrc_pool_copy_to_slot" );
    return Vector3f( slot[0], slot[1], slot[2] );
}

void HipTexturePool::releaseSlot( const Vector3f& pos )
{
    const float slot[3] = { pos[0], pos[1], pos[2] };
    vrc_pool_release_slot( _pool, slot );
}

size_t HipTexturePool::getSlotMemSize() const
{
    size_t slotBytes = 0;
    vrc_pool_info( _pool, &slotBytes, nullptr, nullptr, nullptr, nullptr );
    return slotBytes;
}

Vector3ui HipTexturePool::getTextureSize() const
{
    uint32_t dim[3];
    vrc_pool_info( _pool, nullptr, dim, nullptr, nullptr, nullptr );
    return Vector3ui( dim[0], dim[1], dim[2] );
}

size_t HipTexturePool::getTextureMem() const
{
    size_t atlasBytes = 0;
    vrc_pool_info( _pool, nullptr, nullptr, &atlasBytes, nullptr, nullptr );
    return atlasBytes;
}

/* ---- HipTextureObject: CudaTextureObject.cpp:41-125 ----------------------------------------- */
HipTextureObject::HipTextureObject( const CacheId& cacheId, const DataCache& dataCache,
                                    const DataSource& dataSource, HipTexturePool& pool )
    : CacheObject( cacheId ), _size( 0 ), _texturePool( pool ), _slotPosition( INVALID_SLOT_POSITION )
{
    /* CudaTextureObject.cpp:61-84 */
    const ConstDataObjectPtr data = dataCache.get( cacheId );
    if( !data )
        throw CacheLoadException( cacheId, "Unable to construct texture cache object" );
    const VolumeInformation& volInfo = dataSource.getVolumeInfo();
    _size = _texturePool.getSlotMemSize();
    const LODNode lodNode = dataSource.getNode( NodeId( cacheId ) );
    _worldBox = lodNode.getWorldBox();
    _slotPosition = _texturePool.copyToSlot( static_cast< const uint8_t* >( data->getDataPtr() ),
                                             lodNode.getBlockSize() + volInfo.overlap * 2u );
    if( _slotPosition == INVALID_SLOT_POSITION )
        throw CacheLoadException( cacheId, "Unable to construct texture cache object" );
    const Vector3f cacheTextureSize( _texturePool.getTextureSize() );
    const Vector3f overlap( volInfo.overlap );
    const Vector3f size( lodNode.getVoxelBox().getSize() );
    const Vector3f overlapf = overlap / cacheTextureSize;
    _texturePos = _slotPosition + overlapf;
    _textureSize = size / cacheTextureSize;
}

HipTextureObject::~HipTextureObject()
{
    if( _slotPosition != INVALID_SLOT_POSITION )
        _texturePool.releaseSlot( _slotPosition ); /* CudaTextureObject.cpp:55-59 */
}

/* ---- HipRaycastRenderer: CudaRaycastRenderer.cpp:72-250 -------------------------------------- */
HipRaycastRenderer::HipRaycastRenderer( const std::string& name )
    : RendererPlugin( name ), _ctx( nullptr ), _computedSamplesPerRay( 0 )
{
    checkDeviceLayerAbi();
    throwOnVrcError( vrc_ctx_create( getHipDevice(), &_ctx ), "vrc_ctx_create" );
}

HipRaycastRenderer::~HipRaycastRenderer() { vrc_ctx_destroy( _ctx ); }

namespace
{
uint32_t getShaderDataType( const VolumeInformation& volInfo ) /* CudaRaycastRenderer.cpp:87-105 */
{
    switch( volInfo.dataType )
    {
    case DT_UINT8: case DT_UINT16: case DT_UINT32: return SH_UINT;
    case DT_FLOAT: return SH_FLOAT;
    case DT_INT8: case DT_INT16: case DT_INT32: return SH_INT;
    case DT_UNDEFINED:
    default: throw std::runtime_error( "Unsupported type in the shader." );
    }
}

vrc_view_data makeViewData( const RenderInputs& renderInputs ) /* CudaRaycastRenderer.cpp:136-150 */
{
    const VolumeInformation& volInfo = renderInputs.dataSource.getVolumeInfo();
    const Vector3f halfWorldSize = volInfo.worldSize / 2.0f;
    const Frustum& frustum = renderInputs.frameInfo.frustum;
    vrc_view_data v;
    for( int i = 0; i < 3; ++i )
    {
        v.eyePosition[i] = frustum.getEyePos()[i];
        v.aabbMin[i] = -halfWorldSize[i];
        v.aabbMax[i] = halfWorldSize[i];
    }
    /* glGetIntegerv( GL_VIEWPORT ) in the reference: the channel's pixel viewport */
    for( int i = 0; i < 4; ++i )
        v.glViewport[i] = uint32_t( renderInputs.pixelViewPort[i] );
    for( int i = 0; i < 16; ++i )
    {
        v.invProjMatrix[i] = frustum.getInvProjMatrix().array[i];
        v.modelViewMatrix[i] = frustum.getMVMatrix().array[i];
        v.invViewMatrix[i] = frustum.getInvMVMatrix().array[i];
    }
    v.nearPlane = frustum.nearPlane();
    return v;
}
}

/* update(): cuda/Renderer.cu:245-250 -- transfer function and clip planes to the device layer */
void HipRaycastRenderer::uploadSettings( const RenderInputs& renderInputs )
{
    const std::vector< Vector4f >& planes = renderInputs.renderSettings.getClipPlanes().getPlanes();
    if( planes.size() > 6 )
        throw std::runtime_error( "More than 6 clip planes" );
    float flat[24];
    for( size_t i = 0; i < planes.size(); ++i )
        for( int k = 0; k < 4; ++k )
            flat[i * 4 + k] = planes[i][k];
    int64_t variant = VRC_VARIANT_CUDARAYCASTER;
    (void)vrc_get_option( _ctx, VRC_OPT_VARIANT, &variant );
    std::vector< float > colors = renderInputs.renderSettings.getColorMap().sampleColors();
    if( variant == VRC_VARIANT_GLRAYCASTER )
        /* the GL renderer's transfer function is an RGBA8 texture (GLRaycastRenderer.cpp:188-192) */
        for( float& c : colors )
            c = std::floor( std::min( std::max( c, 0.0f ), 1.0f ) * 255.0f + 0.5f ) / 255.0f;
    throwOnVrcError( vrc_update( _ctx, colors.data(), planes.empty() ? nullptr : flat,
                                 uint32_t( planes.size() ) ),
                     "vrc_update" );
    _uploadedPlanes = planes;
}

void HipRaycastRenderer::preRender( const RenderInputs& renderInputs, const ConstCacheObjects& renderData )
{
    const StageClock clock( StageClock::PreRender );
    uploadSettings( renderInputs );

    const VolumeInformation& volInfo = renderInputs.dataSource.getVolumeInfo();
    if( renderInputs.vrParameters.getSamplesPerRay() == 0 ) /* CudaRaycastRenderer.cpp:113-129 */
    {
        uint32_t maxLOD = 0;
        for( const auto& rb : renderData )
        {
            const uint32_t level = NodeId( rb->getId() ).getLevel();
            if( level > maxLOD )
                maxLOD = level;
        }
        const float maxVoxelDim = float( volInfo.voxels.find_max() );
        const float maxVoxelsAtLOD =
            maxVoxelDim / float( 1u << ( volInfo.rootNode.getDepth() - maxLOD - 1 ) );
        _computedSamplesPerRay = uint32_t( std::max( maxVoxelsAtLOD, float( minSamplesPerRay ) ) );
    }
    else /* the reference leaves the member uninitialised here (quirk Q7); the GL twin uses the flag */
        _computedSamplesPerRay = renderInputs.vrParameters.getSamplesPerRay();

    throwOnVrcError( vrc_set_row_map( _ctx, renderInputs.rowMap.empty() ? nullptr : renderInputs.rowMap.data(),
                                      uint32_t( renderInputs.rowMap.size() ) ),
                     "vrc_set_row_map" );
    const vrc_view_data viewData = makeViewData( renderInputs );
    throwOnVrcError( vrc_pre_render( _ctx, &viewData ), "vrc_pre_render" );
}

void HipRaycastRenderer::render( const RenderInputs& renderInputs, const ConstCacheObjects& renderData )
{
    _lastRayLod = false;
    if( renderData.empty() ) /* CudaRaycastRenderer.cpp:157-158 */
        return;
    /* the passes of one frame may carry different clip planes (per-ray LOD in slabs, HipRaycastPipeline) */
    if( renderInputs.renderSettings.getClipPlanes().getPlanes() != _uploadedPlanes )
        uploadSettings( renderInputs );
    std::unique_ptr< StageClock > clock( new StageClock( StageClock::SortAndFill ) );
    /* CudaRaycastRenderer.cpp:160-163: sort front to back by distance of the box centre.
     * (keys are precomputed; the reference recomputes them inside the comparator) */
    const Frustum& frustum = renderInputs.frameInfo.frustum;
    /* the same bricks seen through the same model-view matrix sort into the same list: a frame
     * that repeats the last one (a standing camera, the ranks of a sort-first frame at a high
     * frame rate) skips the 512 transforms, the sort and the fill (~30 us of host time) */
    bool sameBricks = _sortedFor.size() == renderData.size();
    for( size_t i = 0; sameBricks && i < renderData.size(); ++i )
    {
        /* same object: same address, and -- an address can be recycled -- same brick in the same slot */
        const HipTextureObject& obj = static_cast< const HipTextureObject& >( *renderData[i] );
        sameBricks = _sortedFor[i] == renderData[i].get() && _sortedForIds[i] == obj.getId() &&
                     _sortedForTex[i] == obj.getTexPosition();
    }
    bool sameList = sameBricks && _sortedMV == frustum.getMVMatrix();
    int64_t kernelWanted = VRC_KERNEL_AUTO;
    (void)vrc_get_option( _ctx, VRC_OPT_KERNEL, &kernelWanted );
    const bool orderFree = _orderFree && kernelWanted != VRC_KERNEL_REFERENCE_ORDER;
    if( sameList && !_orderExact && !orderFree )
        sameList = false; /* the kept list is in an older view's order and this kernel marches in list order */
    if( sameBricks && !sameList && orderFree )
    {
        /* the same bricks from another view point, and the last frame's kernel found its bricks
         * through the brick grid (whether the bricks form a grid does not depend on the camera):
         * the order of the node list means nothing to it, so the list stays -- no 512 transforms,
         * no sort, and vrc_render finds the device copy of the table current */
        sameList = true;
        _sortedMV = frustum.getMVMatrix();
        _orderExact = false;
    }
    /* (distance, index into renderData): sorting indices moves no reference counts */
    std::vector< std::pair< float, uint32_t > > keyed;
    if( !sameList )
    {
        keyed.reserve( renderData.size() );
        for( uint32_t i = 0; i < renderData.size(); ++i )
        {
            const Boxf& box = static_cast< const HipTextureObject& >( *renderData[i] ).getWorldBox();
            keyed.push_back( { ( frustum.getMVMatrix() * box.getCenter() ).length(), i } );
        }
    }
    if( !sameList && sameBricks )
    {
        /* the camera moved a little: the last frame's order is still THE stable sort of the new
         * distances if it is non-decreasing with ties in index order -- then the node table stays
         * as it is (and vrc_render finds its device copy current: no rebuild, no upload) */
        bool stillSorted = true;
        for( size_t k = 1; stillSorted && k < _sortedOrder.size(); ++k )
        {
            const float d0 = keyed[_sortedOrder[k - 1]].first, d1 = keyed[_sortedOrder[k]].first;
            stillSorted = d0 < d1 || ( d0 == d1 && _sortedOrder[k - 1] < _sortedOrder[k] );
        }
        if( stillSorted )
        {
            sameList = true;
            _sortedMV = frustum.getMVMatrix();
            _orderExact = true;
        }
        if( StageClock::enabled() )
            ++( stillSorted ? StageClock::orderKept() : StageClock::orderSorted() );
    }
    if( !sameList )
    {
    std::stable_sort( keyed.begin(), keyed.end(),
                      []( const std::pair< float, uint32_t >& a, const std::pair< float, uint32_t >& b ) {
                          return a.first < b.first;
                      } );
    }

    const VolumeInformation& volInfo = renderInputs.dataSource.getVolumeInfo();
    std::vector< vrc_node_data >& nodeDatas = _sortedNodes;
    vrc_pool* pool = _sortedPool;
    if( !sameList )
    {
    nodeDatas.clear();
    nodeDatas.reserve( keyed.size() );
    _sortedIds.clear();
    _sortedOrder.clear();
    pool = nullptr;
    for( const auto& kv : keyed )
    {
        _sortedOrder.push_back( kv.second );
        const HipTextureObject* hipObject = static_cast< const HipTextureObject* >( renderData[kv.second].get() );
        const Boxf& aabb = hipObject->getWorldBox();
        vrc_node_data nd;
        const Vector3f tp = hipObject->getTexPosition(), ts = hipObject->getTexSize(),
                       mn = aabb.getMin(), sz = aabb.getSize();
        for( int i = 0; i < 3; ++i )
        {
            nd.textureMin[i] = tp[i];
            nd.textureSize[i] = ts[i];
            nd.aabbMin[i] = mn[i];
            nd.aabbSize[i] = sz[i];
        }
        nodeDatas.push_back( nd );
        _sortedIds.push_back( hipObject->getId() );
        if( !pool )
            pool = hipObject->getTexturePool()._getHipTexturePool();
    }
    _sortedPool = pool;
    _sortedMV = frustum.getMVMatrix();
    _orderExact = true;
    _sortedFor.resize( renderData.size() );
    _sortedForIds.resize( renderData.size() );
    _sortedForTex.resize( renderData.size() );
    for( size_t i = 0; i < renderData.size(); ++i )
    {
        _sortedFor[i] = renderData[i].get();
        _sortedForIds[i] = renderData[i]->getId();
        _sortedForTex[i] = static_cast< const HipTextureObject& >( *renderData[i] ).getTexPosition();
    }
    }
    clock.reset( new StageClock( StageClock::DeviceCalls ) );
    const vrc_view_data viewData = makeViewData( renderInputs );
    vrc_render_data rData; /* CudaRaycastRenderer.cpp:199-206 */
    rData.samplesPerRay = _computedSamplesPerRay;
    rData.samplesPerPixel = renderInputs.vrParameters.getSamplesPerPixel();
    rData.maxSamplesPerRay = maxSamplesPerRay;
    rData.datatype = getShaderDataType( volInfo );
    rData.dataSourceRange[0] = 0.0f; /* hard-coded (0,255), CudaRaycastRenderer.cpp:205 */
    rData.dataSourceRange[1] = 255.0f;
    if( volInfo.getBytesPerVoxel() != 1 )
    {
        /* extension (16-bit voxels): the range the GL twin takes from the render inputs,
         * GLRaycastRenderer.cpp:311-312 */
        rData.dataSourceRange[0] = renderInputs.dataSourceRange[0];
        rData.dataSourceRange[1] = renderInputs.dataSourceRange[1];
    }
    /* per-ray LOD (extension): SelectVisibles.cpp:55-57 worldSpacePerPixel of this frame */
    const bool rayLod = renderInputs.vrParameters.getRayLOD();
    throwOnVrcError( vrc_set_ray_lod( _ctx, rayLod ? 1 : 0, renderInputs.vrParameters.getSSE(),
                                      ( frustum.top() - frustum.bottom() ) /
                                          float( renderInputs.pixelViewPort[3] ) ),
                     "vrc_set_ray_lod" );
    int rc = vrc_render( _ctx, &viewData, nodeDatas.data(), uint32_t( nodeDatas.size() ), &rData, pool );
    _lastRayLod = rayLod && rc == VRC_OK;
    if( rayLod && rc == VRC_EHIERARCHY )
    {
        /* the bricks do not form level grids (a tree the table builder does not know): render the
         * per-brick cut, i.e. the list without the ancestors the pipeline added */
        std::unordered_set< Identifier > parents;
        for( const Identifier id : _sortedIds )
            for( const NodeId& parent : NodeId( id ).getParents() )
                parents.insert( parent.getId() );
        std::vector< vrc_node_data > cut;
        for( size_t i = 0; i < _sortedIds.size(); ++i )
            if( !parents.count( _sortedIds[i] ) )
                cut.push_back( nodeDatas[i] );
        throwOnVrcError( vrc_set_ray_lod( _ctx, 0, 1.0f, 1.0f ), "vrc_set_ray_lod" );
        rc = vrc_render( _ctx, &viewData, cut.data(), uint32_t( cut.size() ), &rData, pool );
    }
    throwOnVrcError( rc, "vrc_render" );
    int64_t used = VRC_KERNEL_REFERENCE_ORDER;
    (void)vrc_get_option( _ctx, VRC_OPT_KERNEL_USED, &used );
    _orderFree = used != VRC_KERNEL_REFERENCE_ORDER && used != VRC_KERNEL_AUTO;
}

void HipRaycastRenderer::postRender( const RenderInputs&, const ConstCacheObjects& )
{
    /* the reference unmaps the PBO and draws it (cuda/Renderer.cu:299-326); headless: the frame
     * stays in device memory until readFrame / the tile gather takes it */
    throwOnVrcError( vrc_post_render( _ctx, nullptr ), "vrc_post_render" );
}

void HipRaycastRenderer::readFrame( float* hostRgba )
{
    throwOnVrcError( vrc_post_render( _ctx, hostRgba ), "vrc_post_render" );
}
void HipRaycastRenderer::getFrameBuffer( void** d, uint32_t* w, uint32_t* h )
{
    throwOnVrcError( vrc_get_framebuffer( _ctx, d, w, h ), "vrc_get_framebuffer" );
}
void HipRaycastRenderer::setFrameBuffer( void* d, uint32_t w, uint32_t h )
{
    throwOnVrcError( vrc_set_framebuffer( _ctx, d, w, h ), "vrc_set_framebuffer" );
}
void HipRaycastRenderer::setStream( void* s ) { throwOnVrcError( vrc_ctx_set_stream( _ctx, s ), "vrc_ctx_set_stream" ); }
void HipRaycastRenderer::setOption( int o, int64_t v ) { throwOnVrcError( vrc_set_option( _ctx, o, v ), "vrc_set_option" ); }
void HipRaycastRenderer::synchronize() { throwOnVrcError( vrc_synchronize( _ctx ), "vrc_synchronize" ); }
void HipRaycastRenderer::kernelStats( float* lastMs, double* sumMs, uint32_t* launches, uint64_t* samples )
{
    vrc_stats st;
    throwOnVrcError( vrc_get_stats( _ctx, &st ), "vrc_get_stats" );
    if( lastMs ) *lastMs = st.kernel_ms;
    if( sumMs ) *sumMs = st.kernel_ms_sum;
    if( launches ) *launches = st.kernel_launches;
    if( samples ) *samples = st.samples;
}

/* ---- rendering set: RenderingSetGeneratorFilter.ipp:39-95 ----------------------------------- */
ConstCacheObjects generateRenderingSet( const HipTextureCache& cache, const NodeIds& visibles,
                                        RenderStatistics& availability )
{
    ConstCacheMap cacheMap;
    for( const NodeId& nodeId : visibles )
    {
        /* collectLoadedData: the node itself or its nearest cached ancestor */
        NodeId current = nodeId;
        while( current.isValid() )
        {
            const ConstCacheObjectPtr data = cache.get( current.getId() );
            if( data )
            {
                cacheMap[current.getId()] = data;
                break;
            }
            current = current.isRoot() ? NodeId() : current.getParent();
        }
        cacheMap.count( nodeId.getId() ) > 0 ? ++availability.nAvailable : ++availability.nNotAvailable;
    }
    if( visibles.size() != cacheMap.size() )
    {
        /* drop every node that has an ancestor in the map.  (The reference tests this with
         * NodeId::getParents(), RenderingSetGeneratorFilter.ipp:39-48) */
        size_t previousSize = 0;
        do
        {
            previousSize = cacheMap.size();
            for( auto it = cacheMap.begin(); it != cacheMap.end(); )
            {
                bool hasParent = false;
                for( const NodeId& parentId : NodeId( it->first ).getParents() )
                    if( cacheMap.find( parentId.getId() ) != cacheMap.end() )
                    {
                        hasParent = true;
                        break;
                    }
                it = hasParent ? cacheMap.erase( it ) : std::next( it );
            }
        } while( previousSize != cacheMap.size() );
    }
    ConstCacheObjects cacheObjects;
    cacheObjects.reserve( cacheMap.size() );
    for( const auto& kv : cacheMap )
        cacheObjects.push_back( kv.second );
    availability.nRenderAvailable = cacheObjects.size();
    return cacheObjects;
}

/* ---- HipRaycastPipeline: CudaRaycastPipeline.cpp:66-358 -------------------------------------- */
struct HipRaycastPipeline::Impl
{
    Impl()
        : nUploadThreads( configuredUploadThreads() ), _uploadExecutor( nUploadThreads ),
          _asyncUploadExecutor( nAsyncUploadThreads ), _lastPasses( 0 )
    {
    }
    const size_t nUploadThreads;

    /* VisibleSetGeneratorFilter.cpp:42-75 */
    NodeIds visibleSet( const RenderInputs& in ) const
    {
        const RendererParameters& p = in.vrParameters;
        SelectVisibles visitor( in.dataSource, in.frameInfo.frustum, uint32_t( in.pixelViewPort[3] ),
                                p.getSSE(), p.getMinLOD(), p.getMaxLOD(), in.renderDataRange,
                                in.renderSettings.getClipPlanes() );
        DFSTraversal traverser;
        traverser.traverse( in.dataSource.getVolumeInfo().rootNode, visitor, in.frameInfo.timeStep );
        return visitor.getVisibles();
    }

    /* per-ray LOD (extension): the visible set plus every ancestor of it down to minLOD, each id
     * once -- the hierarchy the ray-LOD kernel picks levels from.  false if a node is missing
     * from the tree. */
    bool withAncestors( const RenderInputs& in, NodeIds& ids ) const
    {
        std::unordered_set< Identifier > seen;
        NodeIds all;
        for( const NodeId& id : ids )
        {
            NodeId current = id;
            while( current.isValid() && seen.insert( current.getId() ).second )
            {
                const LODNode node = in.dataSource.getNode( current );
                if( !node.isValid() )
                    return false;
                all.push_back( current );
                if( current.isRoot() || current.getLevel() <= in.vrParameters.getMinLOD() )
                    break;
                current = current.getParent();
            }
        }
        ids.swap( all );
        return true;
    }

    /* DataUploadFilter.cpp:35-49 then CudaTextureUploadFilter.cpp:43-60, for a list of cache
     * misses, on nUploadThreads loaders; element k of the result belongs to ids[k] (empty if
     * the brick could not be made resident) */
    std::vector< ConstCacheObjectPtr > loadParallel( const NodeIds& ids, DataSource& dataSource )
    {
        std::vector< ConstCacheObjectPtr > loaded( ids.size() );
        if( ids.empty() )
            return loaded;
        const size_t perThread = std::max< size_t >( 1, ids.size() / nUploadThreads );
        std::atomic< size_t > pending( 0 );
        std::mutex doneMutex;
        std::condition_variable doneCv;
        std::exception_ptr firstError; /* first failure on a loader thread, rethrown on this one */
        for( size_t i = 0; i < nUploadThreads; ++i )
        {
            const size_t begin = perThread * i;
            if( begin >= ids.size() )
                continue;
            const size_t end = ( i == nUploadThreads - 1 ) ? ids.size() : std::min( begin + perThread, ids.size() );
            ++pending;
            _uploadExecutor.schedule( [&, begin, end] {
                /* Completion is signalled on every way out of the task.  Cache::load only absorbs
                 * CacheLoadException; a reader error (std::runtime_error from the UVF source), an allocation
                 * failure or a failed device call in vrc_pool_copy_to_slot leaves through here too -- the
                 * reference hands such failures to the waiting thread through its futures
                 * (CudaRaycastPipeline.cpp:255-297), a lost notification would park the render thread for good.
                 * Notify under the mutex: the waiter owns the condition variable on its stack and may
                 * destroy it as soon as it can re-acquire the mutex and sees 0 (found by ThreadSanitizer,
                 * tests/host_san/pipeline_stress.cpp). */
                struct Done
                {
                    std::mutex& m;
                    std::condition_variable& cv;
                    std::atomic< size_t >& pending;
                    ~Done()
                    {
                        std::lock_guard< std::mutex > lock( m );
                        --pending;
                        cv.notify_all();
                    }
                } done{ doneMutex, doneCv, pending };
                try
                {
                    double tData = 0.0, tTex = 0.0;
                    const bool trace = std::getenv( "LIVRE_HIP_TRACE" ) != nullptr;
                    for( size_t k = begin; k < end; ++k )
                    {
                        const CacheId id = ids[k].getId();
                        const auto t0 = std::chrono::steady_clock::now();
                        const bool have = bool( _dataCache->load( id, dataSource ) );
                        const auto t1 = std::chrono::steady_clock::now();
                        if( have )
                            loaded[k] = _hipCache->load( id, *_dataCache, dataSource, *_texturePool );
                        if( trace )
                        {
                            tData += std::chrono::duration< double, std::milli >( t1 - t0 ).count();
                            tTex += std::chrono::duration< double, std::milli >( std::chrono::steady_clock::now() - t1 ).count();
                        }
                    }
                    if( trace )
                        std::fprintf( stderr, "[livre_hip] loader: %zu bricks, data source + CPU cache %.1f ms, texture upload %.1f ms\n",
                                      end - begin, tData, tTex );
                }
                catch( ... )
                {
                    std::lock_guard< std::mutex > lock( doneMutex );
                    if( !firstError )
                        firstError = std::current_exception();
                }
            } );
        }
        {
            std::unique_lock< std::mutex > lock( doneMutex );
            doneCv.wait( lock, [&] { return pending == 0; } );
        }
        if( firstError )
            std::rethrow_exception( firstError );
        /* Two uploaders can both pass the cache's "has space" test while only one slot is
         * free (the policy evicts until used < max, i.e. one slot; the reference has the same
         * window, Cache.ipp:132-144 + TexturePool.cu:180-181); the loser's load comes back
         * empty.  Retry those serially: with a single loader the policy always leaves a slot. */
        for( size_t k = 0; k < ids.size(); ++k )
            if( !loaded[k] )
            {
                const CacheId id = ids[k].getId();
                if( _dataCache->load( id, dataSource ) )
                    loaded[k] = _hipCache->load( id, *_dataCache, dataSource, *_texturePool );
            }
        return loaded;
    }

    /* CudaRenderUploadFilter.cpp:57-119: cache hits directly, misses through nUploadThreads
     * data loaders feeding the texture uploader */
    ConstCacheObjects upload( const NodeIds& nodeIds, const RenderInputs& in )
    {
        ConstCacheObjects cacheObjects;
        NodeIds notAvailable;
        std::vector< CacheId > ids;
        ids.reserve( nodeIds.size() );
        for( const NodeId& nodeId : nodeIds )
            ids.push_back( nodeId.getId() );
        std::vector< HipTextureCache::ObjectPtr > resident;
        _hipCache->getMany( ids, resident ); /* one read lock for the frame's texture-cache hits */
        cacheObjects.reserve( nodeIds.size() );
        for( size_t k = 0; k < nodeIds.size(); ++k )
        {
            const NodeId& nodeId = nodeIds[k];
            /* texture-cache hit, or the brick is in the CPU cache and only needs its upload
             * (CudaRenderUploadFilter.cpp:70-84 asks cache.load() for both and lets the
             * constructor throw when the data is missing; a C++ throw costs ~150 us in a
             * process with hundreds of loaded DSOs, so the miss is tested for instead) */
            ConstCacheObjectPtr obj = std::move( resident[k] );
            if( !obj && _dataCache->get( nodeId.getId() ) )
                obj = _hipCache->load( nodeId.getId(), *_dataCache, in.dataSource, *_texturePool );
            if( obj )
                cacheObjects.push_back( obj );
            else
                notAvailable.push_back( nodeId );
        }
        if( notAvailable.empty() )
            return cacheObjects;

        const std::vector< ConstCacheObjectPtr > loaded = loadParallel( notAvailable, in.dataSource );
        for( const auto& obj : loaded )
            if( obj )
                cacheObjects.push_back( obj );
        return cacheObjects;
    }

    /* what the visible set of a frame depends on (visibleSet()); a frame whose key equals the
     * last one's and whose bricks were all resident in one pass is rendered from the kept
     * brick list: no tree traversal, no 512 cache look-ups (~35 us of host time, which is what
     * bounds the frame rate of a rank that renders an eighth of a sort-first frame) */
    struct FrameKey
    {
        Matrix4f mv, proj;
        int32_t windowHeight = 0;
        float sse = 0.f;
        uint32_t minLOD = 0, maxLOD = 0, timeStep = 0;
        Range range{ { 0.f, 0.f } };
        std::vector< Vector4f > planes;
        const DataSource* dataSource = nullptr;
        bool rayLOD = false;
        bool operator==( const FrameKey& o ) const
        {
            if( !( mv == o.mv ) || !( proj == o.proj ) || windowHeight != o.windowHeight || sse != o.sse ||
                minLOD != o.minLOD || maxLOD != o.maxLOD || timeStep != o.timeStep || range != o.range ||
                dataSource != o.dataSource || rayLOD != o.rayLOD || planes.size() != o.planes.size() )
                return false;
            for( size_t i = 0; i < planes.size(); ++i )
                for( int k = 0; k < 4; ++k )
                    if( planes[i][k] != o.planes[i][k] )
                        return false;
            return true;
        }
    };
    static FrameKey frameKey( const RenderInputs& in )
    {
        FrameKey k;
        k.mv = in.frameInfo.frustum.getMVMatrix();
        k.proj = in.frameInfo.frustum.getProjMatrix();
        k.windowHeight = in.pixelViewPort[3];
        k.sse = in.vrParameters.getSSE();
        k.minLOD = in.vrParameters.getMinLOD();
        k.maxLOD = in.vrParameters.getMaxLOD();
        k.timeStep = in.frameInfo.timeStep;
        k.range = in.renderDataRange;
        k.planes = in.renderSettings.getClipPlanes().getPlanes();
        k.dataSource = &in.dataSource;
        k.rayLOD = in.vrParameters.getRayLOD();
        return k;
    }

    /* CudaRaycastPipeline.cpp:129-206 */
    void renderSync( RenderStatistics& statistics, Renderer& renderer, const RenderInputs& in )
    {
        const FrameKey key = frameKey( in );
        if( _keptValid && key == _keptKey )
        {
            RenderInputs plainKept( in );
            plainKept.vrParameters.rayLOD = false;
            renderer.render( _lastRayLod ? in : plainKept, _keptObjects, RENDER_ALL );
            statistics.nAvailable = _keptObjects.size();
            statistics.nNotAvailable = 0;
            statistics.nRenderAvailable = statistics.nAvailable;
            return;
        }
        NodeIds nodeIds;
        {
            const StageClock clock( StageClock::VisibleSet );
            nodeIds = visibleSet( in );
        }
        if( _keptValid && key.rayLOD == _keptKey.rayLOD && key.dataSource == _keptKey.dataSource &&
            key.timeStep == _keptKey.timeStep && key.minLOD == _keptKey.minLOD && nodeIds == _keptVisible )
        {
            /* another view of the same bricks (a camera that moves a little): the brick list is kept */
            _keptKey = key;
            RenderInputs plainKept( in );
            plainKept.vrParameters.rayLOD = false;
            renderer.render( _lastRayLod ? in : plainKept, _keptObjects, RENDER_ALL );
            statistics.nAvailable = _keptObjects.size();
            statistics.nNotAvailable = 0;
            statistics.nRenderAvailable = statistics.nAvailable;
            return;
        }
        /* the kept bricks are referenced, hence not evictable: let go before anything is loaded */
        _keptValid = false;
        _keptObjects.clear();
        const NodeIds visible = nodeIds;
        const uint32_t maxNodesPerPass =
            uint32_t( _texturePool->getTextureMem() / _texturePool->getSlotMemSize() );
        /* per-ray LOD: one pass over the visible set and its ancestors; when that does not fit the
         * atlas (or the tree is ragged) the frame is rendered with the reference's per-brick cut */
        bool rayLod = in.vrParameters.getRayLOD();
        if( rayLod )
        {
            NodeIds hierarchy = nodeIds;
            rayLod = withAncestors( in, hierarchy );
            if( rayLod && hierarchy.size() > maxNodesPerPass )
            {
                /* the hierarchy does not fit the atlas: the reference's answer to "does not fit" is passes
                 * (CudaRaycastPipeline.cpp:149-185); for per-ray LOD a pass is a SLAB of space (below) */
                if( renderRayLodInSlabs( statistics, renderer, in, hierarchy, maxNodesPerPass ) )
                    return;
                rayLod = false;
            }
            if( rayLod )
                nodeIds.swap( hierarchy );
            else if( std::getenv( "LIVRE_HIP_TRACE" ) )
                std::fprintf( stderr, "[livre_hip] per-ray LOD not possible for this frame (ragged tree, or no slab of the atlas's size): per-brick cut\n" );
        }
        RenderInputs plain( in );
        plain.vrParameters.rayLOD = false;
        const RenderInputs& inputs = rayLod ? in : plain;
        _lastRayLod = rayLod;
        const uint32_t numberOfPasses =
            uint32_t( std::ceil( float( nodeIds.size() ) / float( maxNodesPerPass ) ) );
        if( numberOfPasses > 1 )
        {
            /* CudaRaycastPipeline.cpp:146-147: front-to-back order decides which bricks go into
             * which pass.  With a single pass the order is irrelevant here (the renderer sorts
             * the cache objects itself, CudaRaycastRenderer.cpp:160-163), so it is skipped. */
            const Frustum& frustum = in.frameInfo.frustum;
            std::vector< std::pair< float, NodeId > > keyed;
            keyed.reserve( nodeIds.size() );
            for( const NodeId& id : nodeIds )
                keyed.push_back( { nodeDistance( in.dataSource, frustum, id ), id } );
            std::stable_sort( keyed.begin(), keyed.end(),
                              []( const std::pair< float, NodeId >& a, const std::pair< float, NodeId >& b ) {
                                  return a.first < b.first;
                              } );
            for( size_t i = 0; i < keyed.size(); ++i )
                nodeIds[i] = keyed[i].second;
        }
        _lastPasses = numberOfPasses;
        for( uint32_t i = 0; i < numberOfPasses; ++i )
        {
            uint32_t renderStages = RENDER_FRAME;
            if( i == 0 )
                renderStages |= RENDER_BEGIN;
            if( i == numberOfPasses - 1u )
                renderStages |= RENDER_END;
            const size_t startIndex = size_t( i ) * maxNodesPerPass;
            const size_t endIndex = std::min( size_t( i + 1 ) * maxNodesPerPass, nodeIds.size() );
            const NodeIds nodesPerPass( nodeIds.begin() + startIndex, nodeIds.begin() + endIndex );
            /* createAndExecuteSyncPass, CudaRaycastPipeline.cpp:208-234 */
            const auto tU0 = std::chrono::steady_clock::now();
            ConstCacheObjects objects;
            {
                const StageClock clock( StageClock::Upload );
                objects = upload( nodesPerPass, in );
            }
            const auto tU1 = std::chrono::steady_clock::now();
            renderer.render( inputs, objects, renderStages );
            if( numberOfPasses == 1 && objects.size() == nodeIds.size() )
            {
                _keptObjects = objects;
                _keptKey = key;
                _keptVisible = visible;
                _keptValid = true;
            }
            if( std::getenv( "LIVRE_HIP_TRACE" ) )
                std::fprintf( stderr, "[livre_hip] pass %u: upload %.2f ms, render call %.2f ms (%zu bricks)\n", i,
                              std::chrono::duration< double, std::milli >( tU1 - tU0 ).count(),
                              std::chrono::duration< double, std::milli >( std::chrono::steady_clock::now() - tU1 ).count(),
                              objects.size() );
            if( numberOfPasses > 1 )
            {
                /* a multipass frame re-uses the slots: the bricks of this pass must be
                 * evictable before the next pass uploads (the render is stream-ordered
                 * after the uploads, and the next uploads after this render) */
                static_cast< HipRaycastRenderer& >( renderer.getPlugin() ).synchronize();
            }
        }
        if( numberOfPasses == 0 ) /* nothing visible: still begin/end the frame */
            renderer.render( inputs, ConstCacheObjects(), RENDER_BEGIN | RENDER_END );
        statistics.nAvailable = nodeIds.size();
        statistics.nNotAvailable = 0;
        statistics.nRenderAvailable = statistics.nAvailable;
    }

    /* Per-ray LOD over a hierarchy larger than the atlas (round 3).  Per-ray LOD picks, at every point of a ray,
     * the level the screen-space-error rule asks for there and falls back to the next resident one -- so a pass that
     * holds only SOME bricks would make rays fall back where the single pass would not.  A pass is therefore a slab
     * of space across an axis every ray runs along in one direction (below), bounded by faces of the finest bricks, with every brick of every level
     * that reaches into the slab resident, and the rays confined to it by two clip planes (the kernel's own
     * tNearGlobal / tFarGlobal, cuda/Renderer.cu:132-149): inside a slab every ray sees exactly the bricks it would
     * see in the whole hierarchy.  Slabs are rendered front to back into the accumulating pixel buffer
     * (Renderer.cu:151-157), as the reference's passes are.  A run that crosses a slab face is cut there (sampling
     * restarts at the face, as it does at every brick face): the frame is the per-ray LOD frame with those extra
     * restarts, and it is what the oracle renders from the same slabs (tests/test_gpu_parity.py).
     * Returns false (nothing rendered) if even the thinnest slab does not fit, there is no room for two planes, or no
     * axis orders the slabs for every ray (round 4: round 3 took the axis from centre - eye, which composites the
     * rays that run the other way back to front -- eye inside the hierarchy, or outside it on another axis). */
    struct Slab
    {
        float a, b; /* a < b whatever the direction */
        NodeIds ids;
    };

    /* the slabs of a frame, front to back, and the axis they are stacked along; false: none (see above) */
    bool planRayLodSlabs( const RenderInputs& in, const NodeIds& hierarchy, uint32_t maxNodesPerPass,
                          std::vector< Slab >& slabs, int& axisOut ) const
    {
        slabs.clear();
        const std::vector< Vector4f >& userPlanes = in.renderSettings.getClipPlanes().getPlanes();
        if( userPlanes.size() + 2u > 6u || hierarchy.empty() )
            return false;
        const Frustum& frustum = in.frameInfo.frustum;
        std::vector< Boxf > boxes;
        boxes.reserve( hierarchy.size() );
        uint32_t finest = 0;
        Vector3f lo( 1e30f ), hi( -1e30f );
        for( const NodeId& id : hierarchy )
        {
            boxes.push_back( in.dataSource.getNode( id ).getWorldBox() );
            finest = std::max( finest, id.getLevel() );
            for( int a = 0; a < 3; ++a )
            {
                lo[a] = std::min( lo[a], boxes.back().getMin()[a] );
                hi[a] = std::max( hi[a], boxes.back().getMax()[a] );
            }
        }
        /* An axis serves only if EVERY ray that meets the hierarchy runs the same way along it -- a ray that runs the
         * other way would cross a slab face into a slab that has been composited already, back to front.  That holds
         * where the eye lies outside the hierarchy's extent on the axis (a ray that enters [lo, hi] from below goes
         * up), or where the four corner rays of the frustum share a sign along it (every ray of the frustum is a
         * positive combination of them).  Among the axes that serve, the one the view looks along most; none (the eye
         * inside the hierarchy under a wide view): no slabs, the caller renders the per-brick cut. */
        const Vector3f eye = frustum.getEyePos();
        Vector3f corner[4];
        {
            const float xs[2] = { frustum.left(), frustum.right() }, ys[2] = { frustum.bottom(), frustum.top() };
            for( int c = 0; c < 4; ++c )
            {
                const Vector4f d = frustum.getInvMVMatrix() * Vector4f( xs[c & 1], ys[c >> 1], -frustum.nearPlane(), 0.0f );
                corner[c] = Vector3f( d[0], d[1], d[2] );
            }
        }
        const Vector3f view = ( corner[0] + corner[1] + corner[2] + corner[3] ) * 0.25f;
        int axis = -1;
        bool forward = true;
        for( int a = 0; a < 3; ++a )
        {
            int sign = 0; /* +1 / -1: every ray that meets the hierarchy runs up / down the axis */
            if( eye[a] <= lo[a] )
                sign = 1;
            else if( eye[a] >= hi[a] )
                sign = -1;
            else if( corner[0][a] > 0.0f && corner[1][a] > 0.0f && corner[2][a] > 0.0f && corner[3][a] > 0.0f )
                sign = 1;
            else if( corner[0][a] < 0.0f && corner[1][a] < 0.0f && corner[2][a] < 0.0f && corner[3][a] < 0.0f )
                sign = -1;
            if( sign != 0 && ( axis < 0 || std::fabs( view[a] ) > std::fabs( view[axis] ) ) )
            {
                axis = a;
                forward = sign > 0;
            }
        }
        if( axis < 0 )
            return false;
        axisOut = axis;
        /* candidate faces: those of the finest bricks present (coarser faces are among them up to rounding) */
        std::vector< float > faces;
        for( size_t i = 0; i < hierarchy.size(); ++i )
            if( hierarchy[i].getLevel() == finest )
            {
                faces.push_back( boxes[i].getMin()[axis] );
                faces.push_back( boxes[i].getMax()[axis] );
            }
        faces.push_back( lo[axis] );
        faces.push_back( hi[axis] );
        std::sort( faces.begin(), faces.end() );
        const float eps = 1e-5f * std::max( hi[axis] - lo[axis], 1e-6f );
        faces.erase( std::unique( faces.begin(), faces.end(), [eps]( float a, float b ) { return b - a <= eps; } ), faces.end() );
        if( !forward )
            std::reverse( faces.begin(), faces.end() );
        auto inSlab = [&]( size_t i, float a, float b ) { /* the brick reaches into the open slab (a, b), a < b */
            return boxes[i].getMax()[axis] > a + eps && boxes[i].getMin()[axis] < b - eps;
        };
        for( size_t f0 = 0; f0 + 1 < faces.size(); )
        {
            size_t f1 = f0 + 1, best = 0;
            NodeIds bestIds;
            for( ; f1 < faces.size(); ++f1 )
            {
                const float a = std::min( faces[f0], faces[f1] ), b = std::max( faces[f0], faces[f1] );
                NodeIds ids;
                for( size_t i = 0; i < hierarchy.size(); ++i )
                    if( inSlab( i, a, b ) )
                        ids.push_back( hierarchy[i] );
                if( ids.size() > maxNodesPerPass )
                    break;
                best = f1;
                bestIds.swap( ids );
            }
            if( best == 0 )
            {
                slabs.clear();
                return false; /* one layer of the finest bricks with their ancestors is more than the atlas holds */
            }
            if( !bestIds.empty() )
                slabs.push_back( { std::min( faces[f0], faces[best] ), std::max( faces[f0], faces[best] ), bestIds } );
            f0 = best;
        }
        return true;
    }

    /* render the first `count` of the frame's slabs (all of them: the whole frame); returns the bricks rendered */
    size_t renderSlabs( Renderer& renderer, const RenderInputs& in, const std::vector< Slab >& slabs, size_t count, int axis )
    {
        _keptValid = false;
        _keptObjects.clear();
        _lastPasses = uint32_t( count );
        _lastRayLod = true;
        size_t bricks = 0;
        for( size_t i = 0; i < count; ++i )
        {
            uint32_t renderStages = RENDER_FRAME;
            if( i == 0 )
                renderStages |= RENDER_BEGIN;
            if( i + 1 == count )
                renderStages |= RENDER_END;
            RenderInputs slabIn( in );
            Vector4f pa( 0.0f ), pb( 0.0f ); /* kept: n.x + d >= 0 (Renderer.cu:132-146) */
            pa[axis] = 1.0f;
            pa[3] = -slabs[i].a;
            pb[axis] = -1.0f;
            pb[3] = slabs[i].b;
            slabIn.renderSettings.getClipPlanes().addPlane( pa );
            slabIn.renderSettings.getClipPlanes().addPlane( pb );
            ConstCacheObjects objects;
            {
                const StageClock clock( StageClock::Upload );
                objects = upload( slabs[i].ids, in );
            }
            if( objects.size() != slabs[i].ids.size() )
            {
                /* close the frame that slab 0 opened before reporting */
                if( i > 0 )
                    renderer.render( in, ConstCacheObjects(), RENDER_END );
                throw std::runtime_error( "per-ray LOD in slabs: a slab's bricks could not be made resident" );
            }
            renderer.render( slabIn, objects, renderStages );
            _lastRayLod = _lastRayLod && static_cast< HipRaycastRenderer& >( renderer.getPlugin() ).lastRenderUsedRayLOD();
            bricks += objects.size();
            if( std::getenv( "LIVRE_HIP_TRACE" ) )
                std::fprintf( stderr, "[livre_hip] per-ray LOD slab %zu of %zu: axis %d [%g, %g], %zu bricks\n", i + 1,
                              slabs.size(), axis, slabs[i].a, slabs[i].b, objects.size() );
            /* the next slab re-uses slots: this one's bricks must be evictable before it uploads */
            static_cast< HipRaycastRenderer& >( renderer.getPlugin() ).synchronize();
        }
        if( count == 0 )
            renderer.render( in, ConstCacheObjects(), RENDER_BEGIN | RENDER_END );
        return bricks;
    }

    bool renderRayLodInSlabs( RenderStatistics& statistics, Renderer& renderer, const RenderInputs& in,
                              const NodeIds& hierarchy, uint32_t maxNodesPerPass )
    {
        std::vector< Slab > slabs;
        int axis = 0;
        if( !planRayLodSlabs( in, hierarchy, maxNodesPerPass, slabs, axis ) )
            return false;
        const size_t bricks = renderSlabs( renderer, in, slabs, slabs.size(), axis );
        statistics.nAvailable = hierarchy.size();
        statistics.nNotAvailable = 0;
        statistics.nRenderAvailable = bricks;
        return true;
    }

    /* The same in ASYNCHRONOUS mode (round 4; CudaRaycastPipeline.cpp:236-301 renders what is resident and asks for a
     * redraw).  An atlas smaller than the hierarchy cannot keep the frame's bricks from one frame to the next, so
     * "resident" means the CPU data cache here: the frame renders the longest front-to-back PREFIX of slabs whose
     * bricks are all in it (their uploads are host-to-device copies only), the bricks still missing are read in the
     * background, and the redraw filter is told the frame is not complete -- the front of the volume first, the rest
     * slab by slab as the data arrives.  Needs a data cache that holds the hierarchy (else bricks would be evicted
     * before the frame completes: false, the caller renders the per-brick cut). */
    bool renderRayLodInSlabsAsync( RenderStatistics& statistics, Renderer& renderer, const RenderInputs& in,
                                   const NodeIds& hierarchy, uint32_t maxNodesPerPass )
    {
        const VolumeInformation& vi = in.dataSource.getVolumeInfo();
        const size_t brickBytes = size_t( vi.maximumBlockSize[0] ) * vi.maximumBlockSize[1] * vi.maximumBlockSize[2] *
                                  vi.compCount * vi.getBytesPerVoxel();
        if( hierarchy.size() * brickBytes > _dataCache->getStatistics().getMaximumMemory() )
            return false;
        std::vector< Slab > slabs;
        int axis = 0;
        if( !planRayLodSlabs( in, hierarchy, maxNodesPerPass, slabs, axis ) )
            return false;
        /* what is missing, in the order the frame fills in: slab by slab from the front (a brick that reaches into
         * several slabs once) */
        NodeIds missing;
        size_t ready = slabs.size();
        {
            std::unordered_set< uint64_t > seen;
            for( size_t i = 0; i < slabs.size(); ++i )
                for( const NodeId& id : slabs[i].ids )
                    if( !_dataCache->get( id.getId() ) )
                    {
                        ready = std::min( ready, i );
                        if( seen.insert( id.getId() ).second )
                            missing.push_back( id );
                    }
        }
        {
            std::exception_ptr pendingError;
            {
                std::lock_guard< std::mutex > lock( _asyncErrorMutex );
                std::swap( pendingError, _asyncError );
            }
            if( pendingError )
                std::rethrow_exception( pendingError );
        }
        if( !missing.empty() )
        {
            bool expected = false;
            if( _asyncBusy.compare_exchange_strong( expected, true ) )
            {
                DataSource* ds = &in.dataSource;
                _asyncUploadExecutor.schedule( [this, missing, ds] {
                    struct Lower
                    {
                        std::atomic< bool >& flag;
                        ~Lower() { flag = false; }
                    } lower{ _asyncBusy };
                    try
                    {
                        /* the CPU cache only: the atlas is the slabs' (front to back: the order the frame fills in) */
                        for( const NodeId& id : missing )
                            _dataCache->load( id.getId(), *ds );
                    }
                    catch( ... )
                    {
                        std::lock_guard< std::mutex > lock( _asyncErrorMutex );
                        if( !_asyncError )
                            _asyncError = std::current_exception();
                    }
                } );
            }
        }
        const size_t bricks = renderSlabs( renderer, in, slabs, ready, axis );
        statistics.nAvailable = hierarchy.size() - missing.size();
        statistics.nNotAvailable = missing.size();
        statistics.nRenderAvailable = bricks;
        if( in.redrawFilter )
            in.redrawFilter( missing.empty() && ready == slabs.size() );
        return true;
    }

    /* CudaRaycastPipeline.cpp:236-301: render what is resident (or a cached ancestor), upload
     * the visible set in the background, ask for a redraw until everything is available */
    void renderAsync( RenderStatistics& statistics, Renderer& renderer, const RenderInputs& in )
    {
        NodeIds visibles = visibleSet( in );
        ConstCacheObjects objects;
        const uint32_t maxNodes = uint32_t( _texturePool->getTextureMem() / _texturePool->getSlotMemSize() );
        NodeIds hierarchy = visibles;
        bool rayLod = in.vrParameters.getRayLOD() && withAncestors( in, hierarchy );
        if( rayLod && hierarchy.size() > maxNodes )
        {
            /* the hierarchy does not fit the atlas: slabs of space, the prefix whose data has arrived */
            if( renderRayLodInSlabsAsync( statistics, renderer, in, hierarchy, maxNodes ) )
                return;
            rayLod = false;
        }
        _lastRayLod = rayLod;
        if( rayLod )
        {
            /* every resident brick of the hierarchy: the kernel falls back to coarser levels where
             * a brick is still missing, the job generateRenderingSet does for the per-brick cut */
            visibles.swap( hierarchy );
            for( const NodeId& id : visibles )
            {
                const ConstCacheObjectPtr obj = _hipCache->get( id.getId() );
                if( obj )
                    objects.push_back( obj );
                obj ? ++statistics.nAvailable : ++statistics.nNotAvailable;
            }
            statistics.nRenderAvailable = objects.size();
        }
        else
            objects = generateRenderingSet( *_hipCache, visibles, statistics );
        RenderInputs plain( in );
        plain.vrParameters.rayLOD = false;
        const RenderInputs& inputs = rayLod ? in : plain;
        const bool allAvailable = statistics.nNotAvailable == 0;
        {
            /* a failure of the upload pipeline behind an earlier frame surfaces here */
            std::exception_ptr pendingError;
            {
                std::lock_guard< std::mutex > lock( _asyncErrorMutex );
                std::swap( pendingError, _asyncError );
            }
            if( pendingError )
                std::rethrow_exception( pendingError );
        }
        if( !allAvailable )
        {
            /* one upload pipeline in flight at a time on the async executor */
            bool expected = false;
            if( _asyncBusy.compare_exchange_strong( expected, true ) )
            {
                /* the data source outlives the pipeline (it does in the reference: Node owns it);
                 * everything else the task touches is owned by this Impl */
                DataSource* ds = &in.dataSource;
                _asyncUploadExecutor.schedule( [this, visibles, ds] {
                    /* the upload pipeline of the reference's async executor: the misses go
                     * through the same nUploadThreads loaders as a synchronous pass.  The busy flag
                     * is lowered on every way out, and a failed load is kept for the render thread
                     * (asyncError(): the next frame reports it instead of waiting for bricks that
                     * will never arrive) */
                    struct Lower
                    {
                        std::atomic< bool >& flag;
                        ~Lower() { flag = false; }
                    } lower{ _asyncBusy };
                    try
                    {
                        NodeIds missing;
                        for( const NodeId& id : visibles )
                            if( !_hipCache->get( id.getId() ) )
                                missing.push_back( id );
                        loadParallel( missing, *ds );
                    }
                    catch( ... )
                    {
                        std::lock_guard< std::mutex > lock( _asyncErrorMutex );
                        if( !_asyncError )
                            _asyncError = std::current_exception();
                    }
                } );
            }
        }
        renderer.render( inputs, objects, RENDER_ALL );
        if( in.redrawFilter ) /* RedrawFilter, livre/eq/Channel.cpp:64-90 */
            in.redrawFilter( allAvailable );
    }

    /* CudaRaycastPipeline.cpp:303-323 */
    void init( const RenderInputs& in )
    {
        if( _hipCache )
            return;
        std::lock_guard< std::mutex > lock( _initMutex );
        if( _hipCache )
            return;
        const RendererParameters& p = in.vrParameters;
        const size_t gpuMem = size_t( p.getMaxGPUCacheMemoryMB() ) * 1024u * 1024u;
        _texturePool.reset( new HipTexturePool( in.dataSource, gpuMem ) );
        if( !_dataCache )
            _dataCache.reset( new DataCache( "Data Cache", size_t( p.getMaxCPUCacheMemoryMB() ) * 1024u * 1024u ) );
        _hipCache.reset( new HipTextureCache( "TextureCache", _texturePool->getTextureMem() ) );
    }

    ~Impl()
    {
        StageClock::report();
        _asyncUploadExecutor.wait();
        _uploadExecutor.wait();
        /* texture objects release their slots into the pool: drop them before the pool */
        _keptObjects.clear();
        _hipCache.reset();
        _dataCache.reset();
        _texturePool.reset();
    }

    std::unique_ptr< HipTexturePool > _texturePool;
    std::unique_ptr< DataCache > _dataCache;
    std::unique_ptr< HipTextureCache > _hipCache;
    Executor _uploadExecutor, _asyncUploadExecutor;
    std::atomic< bool > _asyncBusy{ false };
    std::mutex _asyncErrorMutex;
    std::exception_ptr _asyncError; /* first failure of an asynchronous upload pipeline, not yet reported */
    std::mutex _initMutex;
    uint32_t _lastPasses;
    bool _lastRayLod = false;
    FrameKey _keptKey;
    ConstCacheObjects _keptObjects;
    NodeIds _keptVisible; /* the visible set (before the ancestors of per-ray LOD) the kept list was made from */
    bool _keptValid = false;
};

HipRaycastPipeline::HipRaycastPipeline( const std::string& name )
    : RenderPipelinePlugin( name ), _impl( new Impl() )
{
}
HipRaycastPipeline::~HipRaycastPipeline() {}

RenderStatistics HipRaycastPipeline::render( Renderer& renderer, const RenderInputs& renderInputs )
{
    if( StageClock::enabled() )
        ++StageClock::frames();
    RenderStatistics statistics;
    _impl->init( renderInputs );
    if( renderInputs.vrParameters.getSynchronousMode() )
        _impl->renderSync( statistics, renderer, renderInputs );
    else
        _impl->renderAsync( statistics, renderer, renderInputs );
    return statistics;
}

const CacheStatistics* HipRaycastPipeline::textureCacheStatistics() const
{
    return _impl->_hipCache ? &_impl->_hipCache->getStatistics() : nullptr;
}
const CacheStatistics* HipRaycastPipeline::dataCacheStatistics() const
{
    return _impl->_dataCache ? &_impl->_dataCache->getStatistics() : nullptr;
}
uint32_t HipRaycastPipeline::lastNumberOfPasses() const { return _impl->_lastPasses; }
bool HipRaycastPipeline::lastFrameUsedRayLOD() const { return _impl->_lastRayLod; }
void HipRaycastPipeline::waitForUploads() { _impl->_asyncUploadExecutor.wait(); }
}

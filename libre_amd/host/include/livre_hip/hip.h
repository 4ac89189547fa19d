/*
 * hip.h -- the MI355X renderer plugin behind Libre's plugin surface: HipTexturePool,
 * HipTextureObject, HipRaycastRenderer (RendererPlugin "hip"), HipRaycastPipeline
 * (RenderPipelinePlugin "hip").  Class for class the counterpart of
 * renderers/cudaRaycaster/{CudaTexturePool,CudaTextureObject,CudaRaycastRenderer,
 * CudaRaycastPipeline}.h; all device work goes through the C ABI in include/vrc_hip.h.
 */
#ifndef LIVRE_HIP_HIP_H
#define LIVRE_HIP_HIP_H

#include "render.h"

#include "vrc_hip.h" /* vrc_node_data */

namespace livre
{
/** Thrown for failed HIP calls, as checkCudaErrors does (cuda/cuda.h:39-53). */
void throwOnVrcError( int rc, const char* what );

/** Process-wide device selection (one process per GPU; the reference hard-codes device 0,
 *  cuda/Renderer.cu:236, quirk Q11). Must be set before the first pool/renderer is created. */
void setHipDevice( int device );
int getHipDevice();

/** renderers/cudaRaycaster/CudaTexturePool.h:34-81 */
class HipTexturePool
{
public:
    HipTexturePool( const DataSource& dataSource, size_t textureMemory );
    ~HipTexturePool();
    /** @return normalized slot origin, or Vector3f(-1) when the pool is full */
    Vector3f copyToSlot( const unsigned char* ptr, const Vector3ui& size );
    void releaseSlot( const Vector3f& pos );
    size_t getSlotMemSize() const;
    Vector3ui getTextureSize() const;
    size_t getTextureMem() const;
    vrc_pool* _getHipTexturePool() const { return _pool; }

private:
    vrc_ctx* _ctx;
    vrc_pool* _pool;
    int _device; /* whose shared pool-creation context _ctx is */
};

/** renderers/cudaRaycaster/CudaTextureObject.h:36-70 */
class HipTextureObject : public CacheObject
{
public:
    /** @throws CacheLoadException when the data cache lacks the brick or no slot is free */
    HipTextureObject( const CacheId& cacheId, const DataCache& dataCache,
                      const DataSource& dataSource, HipTexturePool& pool );
    virtual ~HipTextureObject();
    size_t getSize() const final { return _size; }
    Vector3f getTexPosition() const { return _texturePos; }
    Vector3f getTexSize() const { return _textureSize; }
    HipTexturePool& getTexturePool() const { return _texturePool; }
    /** world box of the brick, kept from construction so the per-frame render path does not
     *  look the node up again (not in the reference, which calls DataSource::getNode) */
    const Boxf& getWorldBox() const { return _worldBox; }

private:
    size_t _size;
    HipTexturePool& _texturePool;
    Vector3f _slotPosition, _texturePos, _textureSize;
    Boxf _worldBox;
};
typedef std::shared_ptr< const HipTextureObject > ConstHipTextureObjectPtr;
typedef Cache< HipTextureObject > HipTextureCache;

/** renderers/cudaRaycaster/CudaRaycastRenderer.h:31-55 */
class HipRaycastRenderer : public RendererPlugin
{
public:
    explicit HipRaycastRenderer( const std::string& name );
    ~HipRaycastRenderer();
    static bool handles( const std::string& name ) { return name == "hip"; }
    void preRender( const RenderInputs& renderInputs, const ConstCacheObjects& renderData ) final;
    void render( const RenderInputs& renderInputs, const ConstCacheObjects& renderData ) final;
    void postRender( const RenderInputs& renderInputs, const ConstCacheObjects& renderData ) final;

    /* headless replacements of the PBO + glDrawPixels tail (cuda/Renderer.cu:299-326) */
    void readFrame( float* hostRgba ); /* W*H*4 floats */
    void getFrameBuffer( void** deviceRgba, uint32_t* width, uint32_t* height );
    void setFrameBuffer( void* deviceRgba, uint32_t width, uint32_t height );
    void setStream( void* hipStream );
    void setOption( int option, int64_t value );
    /** kernel time (HIP events) of the last launch, sum and count since the previous call,
     *  and the sample counter of the last launch; synchronizes the render stream */
    void kernelStats( float* lastMs, double* sumMs, uint32_t* launches, uint64_t* samples );
    void synchronize();
    uint32_t getComputedSamplesPerRay() const { return _computedSamplesPerRay; }
    /** the last render() went through the per-ray LOD kernel (false: per-brick cut) */
    bool lastRenderUsedRayLOD() const { return _lastRayLod; }
    /** ids of the bricks of the last render() in the order the node table was handed to the device layer
     *  (front to back by box-centre distance, CudaRaycastRenderer.cpp:160-163) */
    const std::vector< Identifier >& lastNodeOrder() const { return _sortedIds; }
    /** the device-layer context (C ABI handle): for calls that take it, e.g. the sort-first tile exchange
     *  vrc_comm_create / vrc_gather_tiles, the step eq::Compositor::assembleFrame performs for the
     *  reference (livre/eq/Channel.cpp:519-523) */
    vrc_ctx* deviceContext() const { return _ctx; }

private:
    vrc_ctx* _ctx;
    uint32_t _computedSamplesPerRay;
    bool _lastRayLod = false;
    std::vector< Vector4f > _uploadedPlanes; /* what the device layer has (vrc_update): render() uploads a change */
    void uploadSettings( const RenderInputs& renderInputs );
    /* render(): the sorted node list of the last call, kept while the bricks and the model-view matrix repeat */
    std::vector< const CacheObject* > _sortedFor;
    std::vector< Identifier > _sortedForIds;
    std::vector< Vector3f > _sortedForTex;
    std::vector< Identifier > _sortedIds;
    std::vector< uint32_t > _sortedOrder; /* sorted position -> index in the render data */
    std::vector< vrc_node_data > _sortedNodes;
    Matrix4f _sortedMV;
    vrc_pool* _sortedPool = nullptr;
    bool _orderFree = false; /* the last kernel enumerated bricks through the grid: list order irrelevant */
    bool _orderExact = true; /* the kept node table is in front-to-back order for _sortedMV */
};

/** renderers/cudaRaycaster/CudaRaycastPipeline.h:37-60 */
class HipRaycastPipeline : public RenderPipelinePlugin
{
public:
    explicit HipRaycastPipeline( const std::string& name );
    ~HipRaycastPipeline();
    static bool handles( const std::string& name ) { return name == "hip"; }
    RenderStatistics render( Renderer& renderer, const RenderInputs& renderInputs ) final;

    /* introspection for tests and the bench */
    const CacheStatistics* textureCacheStatistics() const;
    const CacheStatistics* dataCacheStatistics() const;
    uint32_t lastNumberOfPasses() const;
    bool lastFrameUsedRayLOD() const; /* per-ray LOD was possible for the last frame */
    /** block until the asynchronous upload pipeline is idle */
    void waitForUploads();

private:
    struct Impl;
    std::unique_ptr< Impl > _impl;
};

/** livre/lib/pipeline/RenderingSetGeneratorFilter.ipp:39-95: for every visible node take it if
 *  it is in the cache, else its nearest cached ancestor; then drop nodes that have an
 *  ancestor in the set. */
ConstCacheObjects generateRenderingSet( const HipTextureCache& cache, const NodeIds& visibles,
                                        RenderStatistics& availability );
}
#endif

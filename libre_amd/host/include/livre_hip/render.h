/*
 * render.h -- the render-side plugin surface: Frustum, ClipPlanes, settings, RenderInputs,
 * Renderer + RendererPlugin, RenderPipeline + RenderPipelinePlugin, SelectVisibles +
 * DFSTraversal.  Mirrors livre/core/render, livre/core/visitor, livre/core/configuration and
 * livre/core/settings of the reference (signatures and behaviour; citations per declaration).
 */
#ifndef LIVRE_HIP_RENDER_H
#define LIVRE_HIP_RENDER_H

#include "cache.h"

namespace livre
{
/** livre/core/render/Frustum.h:36-104 (vmml::Frustumf limits + FrustumCullerf + matrices) */
class Frustum
{
public:
    Frustum() : Frustum( Matrix4f(), perspectiveFrustum( -0.05f, 0.05f, -0.05f, 0.05f, 0.1f, 15.0f ) ) {}
    Frustum( const Matrix4f& modelViewMatrix, const Matrix4f& projectionMatrix );
    const Plane& getNearPlane() const { return _planes[4]; }
    bool isInFrustum( const Boxf& worldBox ) const;
    const Matrix4f& getMVMatrix() const { return _mvMatrix; }
    const Matrix4f& getProjMatrix() const { return _projMatrix; }
    const Matrix4f& getInvMVMatrix() const { return _invMVMatrix; }
    const Matrix4f& getInvProjMatrix() const { return _invProjMatrix; }
    Matrix4f getMVPMatrix() const { return _projMatrix * _mvMatrix; }
    const Vector3f& getEyePos() const { return _eye; }
    const Vector3f& getViewDir() const { return _dir; }
    float nearPlane() const { return _near; }
    float farPlane() const { return _far; }
    float left() const { return _left; }
    float right() const { return _right; }
    float bottom() const { return _bottom; }
    float top() const { return _top; }
    bool operator==( const Frustum& rhs ) const { return _mvMatrix.equals( rhs._mvMatrix, 1.1920929e-7f ); }

private:
    Matrix4f _mvMatrix, _invMVMatrix, _projMatrix, _invProjMatrix;
    Vector3f _eye, _dir;
    Plane _planes[6]; /* left, right, bottom, top, near, far */
    float _left, _right, _bottom, _top, _near, _far;
};

/** livre/core/render/ClipPlanes.h + ClipPlanes.cpp:70-104; plane = (normal, d), inside is
 *  normal.p + d >= 0.  Default: the six planes of the unit cube at d = 0.5. */
class ClipPlanes
{
public:
    ClipPlanes() { reset(); }
    bool isEmpty() const { return _planes.empty(); }
    void clear() { _planes.clear(); }
    void reset();
    void addPlane( const Vector4f& plane ) { _planes.push_back( plane ); }
    bool isClipped( const Boxf& worldBox ) const;
    const std::vector< Vector4f >& getPlanes() const { return _planes; }

private:
    std::vector< Vector4f > _planes;
};

/** lexis::render::ColorMap reduced to what the renderer consumes: 256 RGBA float samples
 *  (cuda/ColorMap.cu:56-65 calls sampleColors<float>(256, 0, 256, 0)). */
class ColorMap
{
public:
    ColorMap(); /* linear grey ramp (explicit, documented default; Lexis is un-vendored) */
    static ColorMap linearRamp( float alphaScale );
    const std::vector< float >& sampleColors() const { return _rgba; }
    void setSamples( const float* rgba256 ) { _rgba.assign( rgba256, rgba256 + 1024 ); }

private:
    std::vector< float > _rgba;
};

/** livre/core/settings/RenderSettings.h: colour map + clip planes.  The running app starts with
 *  zero planes (livre/eq/settings/EqRenderSettings.cpp:42, quirk Q16). */
class RenderSettings
{
public:
    RenderSettings() { _clipPlanes.clear(); }
    const ColorMap& getColorMap() const { return _colorMap; }
    ColorMap& getColorMap() { return _colorMap; }
    const ClipPlanes& getClipPlanes() const { return _clipPlanes; }
    ClipPlanes& getClipPlanes() { return _clipPlanes; }

private:
    ColorMap _colorMap;
    ClipPlanes _clipPlanes;
};

/** livre/core/configuration/rendererParameters.fbs:4-13 defaults (pinned by
 *  tests/lib/rendererParameters.cpp:25-44) */
struct RendererParameters
{
    uint32_t maxLOD = 9;
    uint32_t minLOD = 0;
    float screenSpaceError = 4.0f;
    bool synchronousMode = false;
    uint32_t samplesPerRay = 0;
    uint32_t samplesPerPixel = 1;
    uint32_t maxGPUCacheMemoryMB = 3072;
    uint32_t maxCPUCacheMemoryMB = 8192;
    /** EXTENSION (not in rendererParameters.fbs): per-ray adaptive LOD.  The pipeline makes the
     *  ancestors of the visible set resident too and the renderer applies the screen-space-error
     *  rule of SelectVisibles along every ray (vrc_set_ray_lod). */
    bool rayLOD = false;
    bool getRayLOD() const { return rayLOD; }
    uint32_t getMaxLOD() const { return maxLOD; }
    uint32_t getMinLOD() const { return minLOD; }
    float getSSE() const { return screenSpaceError; }
    bool getSynchronousMode() const { return synchronousMode; }
    uint32_t getSamplesPerRay() const { return samplesPerRay; }
    uint32_t getSamplesPerPixel() const { return samplesPerPixel; }
    uint32_t getMaxGPUCacheMemoryMB() const { return maxGPUCacheMemoryMB; }
    uint32_t getMaxCPUCacheMemoryMB() const { return maxCPUCacheMemoryMB; }
};

/** livre/core/settings/CameraSettings.cpp:35-103 */
class CameraSettings
{
public:
    void spinModel( float x, float y );
    void moveCamera( float x, float y, float z );
    void setCameraPosition( const Vector3f& pos );
    void setCameraLookAt( const Vector3f& lookAt );
    void setModelViewMatrix( const Matrix4f& mv ) { _modelview = mv; }
    const Matrix4f& getModelViewMatrix() const { return _modelview; }

private:
    Matrix4f _modelview;
};

/** livre/core/render/FrameInfo.h:31-64 */
struct FrameInfo
{
    FrameInfo() : timeStep( 0 ), frameId( 0 ) {}
    FrameInfo( const Frustum& f, uint32_t t, uint32_t id ) : frustum( f ), timeStep( t ), frameId( id ) {}
    Frustum frustum;
    uint32_t timeStep;
    uint32_t frameId;
};

struct RenderStatistics
{
    RenderStatistics() : nAvailable( 0 ), nNotAvailable( 0 ), nRenderAvailable( 0 ) {}
    RenderStatistics& operator+=( const RenderStatistics& na )
    {
        nAvailable += na.nAvailable;
        nNotAvailable += na.nNotAvailable;
        nRenderAvailable += na.nRenderAvailable;
        return *this;
    }
    size_t nAvailable, nNotAvailable, nRenderAvailable;
};

/** livre/core/render/RenderInputs.h:38-50, minus the tuyau filter map (Tuyau is un-vendored
 *  and carries no arithmetic); the three filters the reference passes in (SendHistogram,
 *  Redraw, PreRender) become optional callbacks. */
struct RenderInputs
{
    FrameInfo frameInfo;
    Range renderDataRange;
    Vector2f dataSourceRange;
    PixelViewport pixelViewPort;
    Viewport viewport;
    RenderSettings renderSettings;
    RendererParameters vrParameters;
    std::function< void( bool /*allAvailable*/ ) > redrawFilter; /* livre/eq/Channel.cpp:64-90 */
    DataSource& dataSource;
    /** sort-first row bands (not in the reference, where Equalizer gives a channel one
     *  rectangle): the frame rows this process renders, stacked in its pixel buffer; empty =
     *  the whole pixelViewPort.  pixelViewPort/frustum stay those of the full frame. */
    std::vector< uint32_t > rowMap;
};

/** livre/core/render/Renderer.h:29-32 */
enum RenderStage { RENDER_BEGIN = 1u, RENDER_FRAME = 2u, RENDER_END = 4u, RENDER_ALL = 7u };

/** livre/core/render/RendererPlugin.h:34-72 */
class RendererPlugin
{
public:
    explicit RendererPlugin( const std::string& ) {}
    typedef RendererPlugin PluginT;
    virtual void preRender( const RenderInputs&, const ConstCacheObjects& ) {}
    virtual void render( const RenderInputs&, const ConstCacheObjects& ) = 0;
    virtual void postRender( const RenderInputs&, const ConstCacheObjects& ) {}
    virtual ~RendererPlugin() {}
};

/** livre/core/render/Renderer.h:36-58 + Renderer.cpp:42-54 */
class Renderer
{
public:
    explicit Renderer( const std::string& name );
    ~Renderer();
    void render( const RenderInputs& renderInputs, const ConstCacheObjects& renderData,
                 uint32_t renderStages = RENDER_ALL );
    RendererPlugin& getPlugin() { return *_plugin; }

private:
    std::unique_ptr< RendererPlugin > _plugin;
};

/** livre/core/render/RenderPipelinePlugin.h:31-49 */
class RenderPipelinePlugin
{
public:
    explicit RenderPipelinePlugin( const std::string& ) {}
    typedef RenderPipelinePlugin PluginT;
    virtual RenderStatistics render( Renderer& renderer, const RenderInputs& renderInputs ) = 0;
    virtual ~RenderPipelinePlugin() {}
};

/** livre/core/render/RenderPipeline.h + RenderPipeline.cpp:44-80 */
class RenderPipeline
{
public:
    explicit RenderPipeline( const std::string& name );
    ~RenderPipeline();
    RenderStatistics render( const RenderInputs& renderInputs );
    Renderer& getRenderer() { return *_renderer; }
    RenderPipelinePlugin& getPlugin() { return *_plugin; }

private:
    std::unique_ptr< RenderPipelinePlugin > _plugin;
    std::unique_ptr< Renderer > _renderer;
};

/** livre/core/visitor: VisitState, NodeVisitor, DFSTraversal (DFSTraversal.cpp:33-103) */
class VisitState
{
public:
    VisitState() : _visitChild( true ), _visitNeighbours( true ), _breakTraversal( false ) {}
    bool getVisitChild() const { return _visitChild; }
    bool getVisitNeighbours() const { return _visitNeighbours; }
    bool getBreakTraversal() const { return _breakTraversal; }
    void setVisitChild( bool v ) { _visitChild = v; }
    void setVisitNeighbours( bool v ) { _visitNeighbours = v; }
    void setBreakTraversal( bool v ) { _breakTraversal = v; }

private:
    bool _visitChild, _visitNeighbours, _breakTraversal;
};

class NodeVisitor
{
public:
    virtual ~NodeVisitor() {}
    virtual void visitPre() {}
    virtual void visit( const NodeId& nodeId, VisitState& state ) = 0;
    virtual void visitPost() {}
};

class DFSTraversal
{
public:
    void traverse( const RootNode& rootNode, NodeVisitor& visitor, uint32_t timeStep );

private:
    bool traverse( const NodeId& nodeId, uint32_t depth, NodeVisitor& visitor );
    VisitState _state;
};

/** livre/core/render/SelectVisibles.cpp:33-150: view-dependent LOD cut (screen-space error),
 *  frustum + clip-plane culling; pinned by the golden id lists of tests/lib/lodSelection.cpp */
class SelectVisibles : public NodeVisitor
{
public:
    SelectVisibles( const DataSource& dataSource, const Frustum& frustum, uint32_t windowHeight,
                    float screenSpaceError, uint32_t minLOD, uint32_t maxLOD, const Range& range,
                    const ClipPlanes& clipPlanes );
    void visitPre() final { _visibles.clear(); }
    void visit( const NodeId& nodeId, VisitState& state ) final;
    void visitPost() final;
    const NodeIds& getVisibles() const { return _visibles; }

private:
    bool isLODVisible( const Vector3f& worldCoord, float worldSpacePerVoxel ) const;
    const DataSource& _dataSource;
    const Frustum _frustum;
    const uint32_t _windowHeight;
    const float _screenSpaceError;
    const uint32_t _minLOD, _maxLOD;
    const Range _range;
    const ClipPlanes _clipPlanes;
    NodeIds _visibles;
};
}
#endif

/* render.cpp -- Frustum, ClipPlanes, ColorMap, CameraSettings, Renderer, RenderPipeline,
 * DFSTraversal, SelectVisibles, DataObject.  Mirrors the reference files cited per function. */
#include "livre_hip/render.h"

namespace livre
{
/* ---- DataObject: livre/lib/cache/DataObject.cpp:30-75 ------------------------------------- */
DataObject::DataObject( const CacheId& cacheId, DataSource& dataSource ) : CacheObject( cacheId )
{
    if( dataSource.getVolumeInfo().dataType == DT_UNDEFINED )
        throw std::runtime_error( "Undefined data type" );
    _data = dataSource.getData( NodeId( cacheId ) );
    if( !_data )
        throw CacheLoadException( cacheId, "Unable to construct data cache object" );
}

/* ---- Frustum: livre/core/render/Frustum.cpp:27-43 ------------------------------------------ */
Frustum::Frustum( const Matrix4f& modelViewMatrix, const Matrix4f& projectionMatrix )
    : _mvMatrix( modelViewMatrix ), _invMVMatrix( modelViewMatrix.inverse() ),
      _projMatrix( projectionMatrix ), _invProjMatrix( projectionMatrix.inverse() )
{
    /* vmml::Frustum(projection): limits of a perspective matrix */
    const Matrix4f& m = _projMatrix;
    _near = m( 2, 3 ) / ( m( 2, 2 ) - 1.0f );
    _far = m( 2, 3 ) / ( m( 2, 2 ) + 1.0f );
    _left = _near * ( m( 0, 2 ) - 1.0f ) / m( 0, 0 );
    _right = _near * ( m( 0, 2 ) + 1.0f ) / m( 0, 0 );
    _bottom = _near * ( m( 1, 2 ) - 1.0f ) / m( 1, 1 );
    _top = _near * ( m( 1, 2 ) + 1.0f ) / m( 1, 1 );

    _eye = _invMVMatrix.getTranslation();
    const Vector4f eyeDir = _invMVMatrix.getColumn( 2 );
    _dir = Vector3f( eyeDir[0], eyeDir[1], eyeDir[2] );

    /* vmml::FrustumCuller(proj * mv): planes from the rows of the combined matrix, normalized */
    const Matrix4f pmv = _projMatrix * _mvMatrix;
    const auto row = [&]( size_t r ) { return Vector4f( pmv( r, 0 ), pmv( r, 1 ), pmv( r, 2 ), pmv( r, 3 ) ); };
    const Vector4f r0 = row( 0 ), r1 = row( 1 ), r2 = row( 2 ), r3 = row( 3 );
    const Vector4f raw[6] = { r3 + r0, r3 - r0, r3 + r1, r3 - r1, r3 + r2, r3 - r2 };
    for( int i = 0; i < 6; ++i )
    {
        const float len = std::sqrt( raw[i][0] * raw[i][0] + raw[i][1] * raw[i][1] + raw[i][2] * raw[i][2] );
        _planes[i] = Plane( raw[i][0] / len, raw[i][1] / len, raw[i][2] / len, raw[i][3] / len );
    }
}

/* vmml::FrustumCuller::test != VISIBILITY_NONE (Frustum.cpp:48-52) */
bool Frustum::isInFrustum( const Boxf& worldBox ) const
{
    const Vector3f middle = worldBox.getCenter();
    const Vector3f extent = worldBox.getSize() * 0.5f;
    for( int i = 0; i < 6; ++i )
    {
        const Plane& p = _planes[i];
        const float d = p.dot( middle );
        const float n = extent[0] * std::fabs( p.a ) + extent[1] * std::fabs( p.b ) + extent[2] * std::fabs( p.c );
        if( d - n >= 0 )
            continue; /* fully on the inside of this plane */
        if( d + n > 0 )
            continue; /* intersecting */
        return false;
    }
    return true;
}

/* ---- ClipPlanes: livre/core/render/ClipPlanes.cpp:25-104 ----------------------------------- */
void ClipPlanes::reset()
{
    static const float normals[6][3] = { { -1.0f, 0.0f, 0.0f }, { 1.0f, 0.0f, 0.0f }, { 0.0f, -1.0f, 0.0f },
                                         { 0.0f, 1.0f, 0.0f },  { 0.0f, 0.0f, -1.0f }, { 0.0f, 0.0f, 1.0f } };
    clear();
    for( size_t i = 0; i < 6; ++i )
        _planes.push_back( Vector4f( normals[i][0], normals[i][1], normals[i][2], 0.5f ) );
}

bool ClipPlanes::isClipped( const Boxf& worldBox ) const
{
    for( const Vector4f& v : _planes )
    {
        const Vector3f middle = worldBox.getCenter();
        const Vector3f extent = worldBox.getSize() * 0.5f;
        const Plane plane( v[0], v[1], v[2], v[3] );
        const float d = plane.dot( middle );
        const float n = extent[0] * std::fabs( plane.x() ) + extent[1] * std::fabs( plane.y() ) +
                        extent[2] * std::fabs( plane.z() );
        if( !( d - n >= 0 || d + n > 0 ) )
            return true;
    }
    return false;
}

/* ---- ColorMap -------------------------------------------------------------------------------- */
ColorMap::ColorMap() : _rgba( 1024 )
{
    for( int i = 0; i < 256; ++i )
        _rgba[i * 4 + 0] = _rgba[i * 4 + 1] = _rgba[i * 4 + 2] = _rgba[i * 4 + 3] = float( i ) / 255.0f;
}

/* BASELINE.md TF: rgba[i] = (i/255, i/255, i/255, alpha*i/255) */
ColorMap ColorMap::linearRamp( float alphaScale )
{
    ColorMap c;
    for( int i = 0; i < 256; ++i )
    {
        const float v = float( i ) / 255.0f;
        c._rgba[i * 4 + 0] = c._rgba[i * 4 + 1] = c._rgba[i * 4 + 2] = v;
        c._rgba[i * 4 + 3] = alphaScale * v;
    }
    return c;
}

/* ---- CameraSettings: livre/core/settings/CameraSettings.cpp:35-103 -------------------------- */
void CameraSettings::spinModel( float x, float y )
{
    if( x == 0.f && y == 0.f )
        return;
    Matrix4f mv = _modelview;
    const float t[3] = { mv( 0, 3 ), mv( 1, 3 ), mv( 2, 3 ) };
    mv( 0, 3 ) = mv( 1, 3 ) = mv( 2, 3 ) = 0.0f;
    mv.pre_rotate_x( x );
    mv.pre_rotate_y( y );
    mv( 0, 3 ) = t[0];
    mv( 1, 3 ) = t[1];
    mv( 2, 3 ) = t[2];
    _modelview = mv;
}

void CameraSettings::moveCamera( float x, float y, float z )
{
    _modelview( 0, 3 ) += x;
    _modelview( 1, 3 ) += y;
    _modelview( 2, 3 ) += z;
}

void CameraSettings::setCameraPosition( const Vector3f& pos )
{
    _modelview( 0, 3 ) = pos[0];
    _modelview( 1, 3 ) = pos[1];
    _modelview( 2, 3 ) = pos[2];
}

void CameraSettings::setCameraLookAt( const Vector3f& lookAt )
{
    const Vector3f eye( _modelview( 0, 3 ), _modelview( 1, 3 ), _modelview( 2, 3 ) );
    const Vector3f zAxis = normalize( eye - lookAt );
    Vector3f up( 0.f, 1.f, 0.f );
    const float angle = zAxis.dot( up );
    if( 1.f - std::fabs( angle ) < 0.0001f )
    {
        /* gimbal-lock guard of CameraSettings.cpp:91-99: tilt up by 0.01 rad about +/-x */
        const float s = ( angle > 0 ) ? -1.f : 1.f;
        const float a = 0.01f * s;
        up = Vector3f( 0.f, std::cos( a ), std::sin( a ) );
        up = normalize( up );
    }
    _modelview = Matrix4f( eye, lookAt, up );
}

/* ---- Renderer / RenderPipeline: Renderer.cpp:34-80, RenderPipeline.cpp:34-80 ---------------- */
Renderer::Renderer( const std::string& name )
    : _plugin( PluginFactory< RendererPlugin, const std::string& >::getInstance().create( name ) )
{
}
Renderer::~Renderer() {}

void Renderer::render( const RenderInputs& renderInputs, const ConstCacheObjects& renderData,
                       uint32_t renderStages )
{
    if( renderStages & RENDER_BEGIN )
        _plugin->preRender( renderInputs, renderData );
    if( renderStages & RENDER_FRAME )
        _plugin->render( renderInputs, renderData );
    if( renderStages & RENDER_END )
        _plugin->postRender( renderInputs, renderData );
}

RenderPipeline::RenderPipeline( const std::string& name )
    : _plugin( PluginFactory< RenderPipelinePlugin, const std::string& >::getInstance().create( name ) ),
      _renderer( new Renderer( name ) )
{
}
RenderPipeline::~RenderPipeline() {}

RenderStatistics RenderPipeline::render( const RenderInputs& renderInputs )
{
    return _plugin->render( *_renderer, renderInputs );
}

/* ---- DFSTraversal: livre/core/visitor/DFSTraversal.cpp:33-103 ------------------------------- */
bool DFSTraversal::traverse( const NodeId& nodeId, uint32_t depth, NodeVisitor& visitor )
{
    if( depth == 0 || _state.getBreakTraversal() )
        return false;
    _state = VisitState();
    visitor.visit( nodeId, _state );
    if( _state.getBreakTraversal() || !_state.getVisitChild() )
    {
        _state.setVisitChild( true );
        return false;
    }
    /* children in the order of NodeId::getChildren (x outer, y, z inner), without the vector */
    const Vector3ui childPos = nodeId.getPosition() * 2u;
    bool stop = false;
    for( uint32_t x = 0; x < 2 && !stop; ++x )
        for( uint32_t y = 0; y < 2 && !stop; ++y )
            for( uint32_t z = 0; z < 2 && !stop; ++z )
            {
                traverse( NodeId( nodeId.getLevel() + 1,
                                  Vector3ui( childPos[0] + x, childPos[1] + y, childPos[2] + z ),
                                  nodeId.getTimeStep() ),
                          depth - 1, visitor );
                if( !_state.getVisitNeighbours() )
                    stop = true;
            }
    _state.setVisitNeighbours( true );
    const bool ret = _state.getBreakTraversal();
    _state.setBreakTraversal( false );
    return ret;
}

void DFSTraversal::traverse( const RootNode& rootNode, NodeVisitor& visitor, uint32_t timeStep )
{
    visitor.visitPre();
    const Vector3ui blockSize = rootNode.getBlockSize();
    for( uint32_t x = 0; x < blockSize[0]; ++x )
        for( uint32_t y = 0; y < blockSize[1]; ++y )
            for( uint32_t z = 0; z < blockSize[2]; ++z )
                traverse( NodeId( 0, Vector3ui( x, y, z ), timeStep ), rootNode.getDepth(), visitor );
    visitor.visitPost();
}

/* ---- SelectVisibles: livre/core/render/SelectVisibles.cpp:33-150 ---------------------------- */
SelectVisibles::SelectVisibles( const DataSource& dataSource, const Frustum& frustum,
                                uint32_t windowHeight, float screenSpaceError, uint32_t minLOD,
                                uint32_t maxLOD, const Range& range, const ClipPlanes& clipPlanes )
    : _dataSource( dataSource ), _frustum( frustum ), _windowHeight( windowHeight ),
      _screenSpaceError( screenSpaceError ), _minLOD( minLOD ), _maxLOD( maxLOD ), _range( range ),
      _clipPlanes( clipPlanes )
{
}

/* SelectVisibles.cpp:52-68 */
bool SelectVisibles::isLODVisible( const Vector3f& worldCoord, float worldSpacePerVoxel ) const
{
    const float t = _frustum.top();
    const float b = _frustum.bottom();
    const float worldSpacePerPixel = ( t - b ) / float( _windowHeight );
    const float pixelPerVoxel = worldSpacePerVoxel / worldSpacePerPixel;
    const Vector4f hWorldCoord( worldCoord[0], worldCoord[1], worldCoord[2], 1.0f );
    const float distance = std::fabs( _frustum.getNearPlane().dot( hWorldCoord ) );
    const float n = _frustum.nearPlane();
    const float pixelPerVoxelInDistance = pixelPerVoxel * n / ( n + distance );
    return pixelPerVoxelInDistance <= _screenSpaceError;
}

/* SelectVisibles.cpp:70-113 */
void SelectVisibles::visit( const NodeId& nodeId, VisitState& state )
{
    const LODNode lodNode = _dataSource.getNode( nodeId );
    const Boxf& worldBox = lodNode.getWorldBox();
    if( !_frustum.isInFrustum( worldBox ) || _clipPlanes.isClipped( worldBox ) )
    {
        state.setVisitChild( false );
        return;
    }
    /* vmml::AABB::computeNearFar: the corners nearest / farthest along the plane normal */
    const Plane& nearPlane = _frustum.getNearPlane();
    Vector3f vmin, vmax;
    const float nrm[3] = { nearPlane.a, nearPlane.b, nearPlane.c };
    for( size_t i = 0; i < 3; ++i )
    {
        if( nrm[i] >= 0.0f )
        {
            vmin[i] = worldBox.getMin()[i];
            vmax[i] = worldBox.getMax()[i];
        }
        else
        {
            vmin[i] = worldBox.getMax()[i];
            vmax[i] = worldBox.getMin()[i];
        }
    }
    const Vector4f hVmin( vmin[0], vmin[1], vmin[2], 1.0f ), hVmax( vmax[0], vmax[1], vmax[2], 1.0f );
    if( nearPlane.dot( hVmin ) < 0 || nearPlane.dot( hVmax ) < 0 )
        vmin = _frustum.getEyePos() - _frustum.getViewDir() * _frustum.nearPlane();

    const Vector3ui vb = lodNode.getVoxelBox().getSize();
    const Vector3f voxelBox = Vector3f( vb );
    const Vector3f worldSpacePerVoxel = worldBox.getSize() / voxelBox;
    bool lodVisible = isLODVisible( vmin, worldSpacePerVoxel.find_min() );

    const uint32_t depth = _dataSource.getVolumeInfo().rootNode.getDepth();
    lodVisible = ( lodVisible && lodNode.getRefLevel() >= _minLOD ) ||
                 ( lodNode.getRefLevel() == _maxLOD ) || ( lodNode.getRefLevel() == depth - 1 );
    if( lodVisible )
        _visibles.push_back( lodNode.getNodeId() );
    state.setVisitChild( !lodVisible );
}

/* SelectVisibles.cpp:120-142: sort-last range selection (the slice [range0, range1) of the list) */
void SelectVisibles::visitPost()
{
    const size_t startIndex = size_t( _range[0] * _visibles.size() );
    const size_t endIndex = size_t( _range[1] * _visibles.size() );
    NodeIds selected;
    for( size_t i = 0; i < _visibles.size(); ++i )
        if( i >= startIndex && i < endIndex )
            selected.push_back( _visibles[i] );
    _visibles.swap( selected );
}
}

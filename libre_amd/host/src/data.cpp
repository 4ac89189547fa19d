/* data.cpp -- mathTypes + NodeId + VolumeInformation + LODNode + DataSource plumbing.
 * Mirrors the reference files cited per function (path:line from the reference root). */
#include <sys/mman.h>
#include <cstdlib>
#include <new>
#include "livre_hip/data.h"

#include <algorithm>
#include <cstring>

namespace livre
{
/* ---- AllocMemoryUnit --------------------------------------------------------------------- */
AllocMemoryUnit::AllocMemoryUnit( size_t size ) : _data( nullptr ), _size( size )
{
    const size_t huge = size_t( 2 ) << 20;
    void* p = nullptr;
    if( size >= huge )
    {
        if( posix_memalign( &p, huge, ( size + huge - 1 ) / huge * huge ) != 0 )
            p = nullptr;
#if defined( MADV_HUGEPAGE )
        if( p )
            (void)madvise( p, ( size + huge - 1 ) / huge * huge, MADV_HUGEPAGE );
#endif
    }
    else
        p = std::malloc( size ? size : 1 );
    if( !p )
        throw std::bad_alloc();
    _data = static_cast< uint8_t* >( p );
}

AllocMemoryUnit::~AllocMemoryUnit() { std::free( _data ); }

/* ---- matrices --------------------------------------------------------------------------- */
/* vmmlib Matrix4(eye, lookAt, up) (gluLookAt); pinned by tests/eq/settings/cameraSettings.cpp:99-117 */
Matrix4f::Matrix4f( const Vector3f& eye, const Vector3f& lookAt, const Vector3f& up )
{
    const Vector3f f = normalize( lookAt - eye );
    const Vector3f s = normalize( cross( f, up ) );
    const Vector3f u = cross( s, f );
    *this = Matrix4f();
    for( size_t c = 0; c < 3; ++c )
    {
        ( *this )( 0, c ) = s[c];
        ( *this )( 1, c ) = u[c];
        ( *this )( 2, c ) = -f[c];
    }
    ( *this )( 0, 3 ) = -s.dot( eye );
    ( *this )( 1, 3 ) = -u.dot( eye );
    ( *this )( 2, 3 ) = f.dot( eye );
}

Matrix4f Matrix4f::operator*( const Matrix4f& o ) const
{
    Matrix4f r;
    for( size_t c = 0; c < 4; ++c )
        for( size_t row = 0; row < 4; ++row )
        {
            float s = 0.f;
            for( size_t k = 0; k < 4; ++k )
                s += ( *this )( row, k ) * o( k, c );
            r( row, c ) = s;
        }
    return r;
}

Vector4f Matrix4f::operator*( const Vector4f& v ) const
{
    Vector4f r;
    for( size_t row = 0; row < 4; ++row )
        r[row] = ( *this )( row, 0 ) * v[0] + ( *this )( row, 1 ) * v[1] +
                 ( *this )( row, 2 ) * v[2] + ( *this )( row, 3 ) * v[3];
    return r;
}

Vector3f Matrix4f::operator*( const Vector3f& v ) const
{
    const Vector4f r = ( *this ) * Vector4f( v[0], v[1], v[2], 1.0f );
    return Vector3f( r[0] / r[3], r[1] / r[3], r[2] / r[3] );
}

/* cofactor inverse evaluated in double, rounded once (vmmlib Matrix4::inverse, call sites
 * livre/core/render/Frustum.cpp:31,34) */
Matrix4f Matrix4f::inverse() const
{
    double m[16], inv[16];
    for( int i = 0; i < 16; ++i ) m[i] = array[i];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    Matrix4f r;
    if( det == 0.0 )
        return r;
    for( int i = 0; i < 16; ++i )
        r.array[i] = float( inv[i] / det );
    return r;
}

/* rotation sign convention of vmmlib pre_rotate_x/y, pinned by
 * tests/eq/settings/cameraSettings.cpp:44-57 */
void Matrix4f::pre_rotate_x( float angle )
{
    Matrix4f r;
    const float c = std::cos( angle ), s = std::sin( angle );
    r( 1, 1 ) = c;  r( 1, 2 ) = s;
    r( 2, 1 ) = -s; r( 2, 2 ) = c;
    *this = r * ( *this );
}

void Matrix4f::pre_rotate_y( float angle )
{
    Matrix4f r;
    const float c = std::cos( angle ), s = std::sin( angle );
    r( 0, 0 ) = c; r( 0, 2 ) = -s;
    r( 2, 0 ) = s; r( 2, 2 ) = c;
    *this = r * ( *this );
}

Matrix4f perspectiveFrustum( float l, float r, float b, float t, float n, float f )
{
    Matrix4f m;
    std::memset( m.array, 0, sizeof( m.array ) );
    m( 0, 0 ) = 2.f * n / ( r - l );
    m( 1, 1 ) = 2.f * n / ( t - b );
    m( 0, 2 ) = ( r + l ) / ( r - l );
    m( 1, 2 ) = ( t + b ) / ( t - b );
    m( 2, 2 ) = -( f + n ) / ( f - n );
    m( 3, 2 ) = -1.f;
    m( 2, 3 ) = -2.f * f * n / ( f - n );
    return m;
}

/* ---- NodeId: livre/core/data/NodeId.cpp:33-162 ------------------------------------------ */
NodeId::NodeId( uint32_t level, const Vector3ui& position, uint32_t timeStep )
    : _id( Identifier( level & 0xFu ) | ( Identifier( position[0] & 0x3FFFu ) << 4 ) |
           ( Identifier( position[1] & 0x3FFFu ) << 18 ) |
           ( Identifier( position[2] & 0x3FFFu ) << 32 ) |
           ( Identifier( timeStep & 0x3FFFFu ) << 46 ) )
{
}

Vector3ui NodeId::getPosition() const
{
    return Vector3ui( uint32_t( ( _id >> 4 ) & 0x3FFFu ), uint32_t( ( _id >> 18 ) & 0x3FFFu ),
                      uint32_t( ( _id >> 32 ) & 0x3FFFu ) );
}

NodeIds NodeId::getParents() const
{
    NodeIds nodeIds;
    NodeId parent = getParent();
    while( parent.isValid() )
    {
        nodeIds.push_back( parent );
        parent = parent.getParent();
    }
    return nodeIds;
}

NodeId NodeId::getParent() const
{
    if( getLevel() == INVALID_LEVEL || getLevel() == 0 )
        return NodeId();
    const Vector3ui p = getPosition();
    return NodeId( getLevel() - 1, Vector3ui( p[0] / 2, p[1] / 2, p[2] / 2 ), getTimeStep() );
}

/* NodeId.cpp:70-85, including its shift direction (child position << levelDiff) */
bool NodeId::isParent( const NodeId& parentNodeId ) const
{
    if( parentNodeId.getLevel() >= getLevel() || parentNodeId.getTimeStep() != getTimeStep() ||
        parentNodeId._id == _id )
        return false;
    const uint32_t levelDiff = getLevel() - parentNodeId.getLevel();
    const Vector3ui p = getPosition(), q = parentNodeId.getPosition();
    const Identifier mask = 0x3FFFu;
    return ( ( Identifier( p[0] ) << levelDiff ) & mask ) == q[0] &&
           ( ( Identifier( p[1] ) << levelDiff ) & mask ) == q[1] &&
           ( ( Identifier( p[2] ) << levelDiff ) & mask ) == q[2];
}

NodeIds NodeId::getChildren() const
{
    if( getLevel() == INVALID_LEVEL )
        return NodeIds();
    NodeIds nodeIds;
    const Vector3ui childPos = getPosition() * 2u;
    for( uint32_t x = 0; x < 2; ++x )
        for( uint32_t y = 0; y < 2; ++y )
            for( uint32_t z = 0; z < 2; ++z )
                nodeIds.push_back( NodeId( getLevel() + 1,
                                           Vector3ui( childPos[0] + x, childPos[1] + y, childPos[2] + z ),
                                           getTimeStep() ) );
    return nodeIds;
}

NodeId NodeId::getRoot() const
{
    const Vector3ui p = getPosition();
    const uint32_t d = 1u << getLevel();
    return NodeId( 0, Vector3ui( p[0] / d, p[1] / d, p[2] / d ), getTimeStep() );
}

NodeIds NodeId::getSiblings() const
{
    if( getLevel() == INVALID_LEVEL || getLevel() == 0 )
        return NodeIds();
    return getParent().getChildren();
}

Range NodeId::getRange() const
{
    const size_t width = size_t( 1 ) << getLevel();
    const size_t nChildren = width * width * width;
    const Vector3ui pos = getPosition();
    const size_t position = pos[0] * width * width + pos[1] * width + pos[2];
    const float span = 1.f / float( nChildren );
    const float begin = float( position ) / float( nChildren );
    return Range{ { begin, begin + span } };
}

NodeIds NodeId::getChildrenAtLevel( uint32_t level ) const
{
    if( getLevel() == INVALID_LEVEL || getLevel() >= level )
        return NodeIds();
    NodeIds nodeIds;
    const uint32_t childCount = 1u << ( level - getLevel() );
    const Vector3ui start = getPosition() * childCount;
    for( uint32_t x = 0; x < childCount; ++x )
        for( uint32_t y = 0; y < childCount; ++y )
            for( uint32_t z = 0; z < childCount; ++z )
                nodeIds.push_back( NodeId( level, Vector3ui( start[0] + x, start[1] + y, start[2] + z ),
                                           getTimeStep() ) );
    return nodeIds;
}

/* ---- VolumeInformation: livre/core/data/VolumeInformation.cpp:25-59 --------------------- */
VolumeInformation::VolumeInformation()
    : bigEndian( false ), compCount( 1u ), dataType( DT_UINT8 ), overlap( 0u ),
      maximumBlockSize( 0u ), voxels( 256u ), worldSize( 0.0f ),
      resolution( Vector3f( -1.0f, -1.0f, -1.0f ) ), worldSpacePerVoxel( 0.0f ),
      meterToDataUnitRatio( 1.0f ), frameRange( Vector2ui( INVALID_TIMESTEP ) )
{
}

size_t VolumeInformation::getBytesPerVoxel() const
{
    switch( dataType )
    {
    case DT_FLOAT: case DT_UINT32: case DT_INT32: return 4;
    case DT_UINT16: case DT_INT16: return 2;
    case DT_UINT8: case DT_INT8: return 1;
    default: return size_t( -1 );
    }
}

/* livre/core/data/DataSourcePlugin.cpp:83-109 */
bool fillRegularVolumeInfo( VolumeInformation& info )
{
    info.worldSpacePerVoxel = 1.0f / float( info.voxels.find_max() );
    info.worldSize = Vector3f( float( info.voxels[0] ), float( info.voxels[1] ), float( info.voxels[2] ) ) *
                     info.worldSpacePerVoxel;
    const Vector3ui blockSize = info.maximumBlockSize - info.overlap * 2u;
    Vector3ui numBlocks, lodLevels;
    for( size_t i = 0; i < 3; ++i )
    {
        numBlocks[i] = uint32_t( std::ceil( float( info.voxels[i] ) / blockSize[i] ) );
        lodLevels[i] = uint32_t( std::ceil( std::log2( double( numBlocks[i] ) ) ) );
    }
    const uint32_t depth = lodLevels.find_min();
    Vector3ui rootNodeBlocksCount;
    for( size_t i = 0; i < 3; ++i )
        rootNodeBlocksCount[i] = uint32_t( std::ceil( float( info.voxels[i] >> depth ) / blockSize[i] ) );
    info.rootNode = RootNode( depth + 1, rootNodeBlocksCount );
    return true;
}

/* ---- LODNode: livre/core/data/LODNode.cpp:36-66 ------------------------------------------ */
LODNode::LODNode( const NodeId& nodeId, const Vector3ui& blockSize, const Boxf& worldBox )
    : _nodeId( nodeId ), _blockSize( blockSize ), _worldBox( worldBox )
{
    const Vector3ui pntPos = getAbsolutePosition() * _blockSize;
    _localVoxelBox = Boxui( pntPos, pntPos + _blockSize );
}

/* ---- URI ----------------------------------------------------------------------------------- */
URI::URI( const std::string& str ) : _str( str )
{
    std::string rest = str;
    const size_t hash = rest.find( '#' );
    if( hash != std::string::npos )
    {
        _fragment = rest.substr( hash + 1 );
        rest = rest.substr( 0, hash );
    }
    const size_t q = rest.find( '?' );
    if( q != std::string::npos )
    {
        std::string query = rest.substr( q + 1 );
        rest = rest.substr( 0, q );
        size_t pos = 0;
        while( pos <= query.size() )
        {
            size_t amp = query.find( '&', pos );
            if( amp == std::string::npos ) amp = query.size();
            const std::string kv = query.substr( pos, amp - pos );
            const size_t eq = kv.find( '=' );
            if( !kv.empty() )
                _query.push_back( { kv.substr( 0, eq ), eq == std::string::npos ? "" : kv.substr( eq + 1 ) } );
            pos = amp + 1;
        }
    }
    const size_t sep = rest.find( "://" );
    if( sep != std::string::npos )
    {
        _scheme = rest.substr( 0, sep );
        _path = rest.substr( sep + 3 );
    }
    else
        _path = rest;
}

bool URI::findQuery( const std::string& key, std::string& value ) const
{
    for( const auto& kv : _query )
        if( kv.first == key )
        {
            value = kv.second;
            return true;
        }
    return false;
}

/* ---- DataSourcePlugin: livre/core/data/DataSourcePlugin.cpp:24-81 ------------------------- */
DataSourcePlugin::DataSourcePlugin() : _lodNodeMap( 128 ) {}

LODNode DataSourcePlugin::getNode( const NodeId& nodeId ) const
{
    if( nodeLookupIsCheap() )
        return internalNodeToLODNode( nodeId );
    {
        std::shared_lock< std::shared_timed_mutex > lock( _mutex );
        const auto it = _lodNodeMap.find( nodeId.getId() );
        if( it != _lodNodeMap.end() )
            return it->second;
    }
    std::unique_lock< std::shared_timed_mutex > lock( _mutex );
    auto it = _lodNodeMap.find( nodeId.getId() );
    if( it == _lodNodeMap.end() )
        it = _lodNodeMap.emplace( nodeId.getId(), internalNodeToLODNode( nodeId ) ).first;
    return it->second;
}

LODNode DataSourcePlugin::internalNodeToLODNode( const NodeId& internalNode ) const
{
    const uint32_t refLevel = internalNode.getLevel();
    const Vector3ui bricksInRefLevel = _volumeInfo.rootNode.getBlockSize( refLevel );
    const Vector3ui pos = internalNode.getPosition();
    Vector3f boxCoordMin = Vector3f( pos );
    Vector3f boxCoordMax = Vector3f( pos + 1u );
    const size_t index = bricksInRefLevel.find_max_index();
    boxCoordMin = boxCoordMin / float( bricksInRefLevel[index] );
    boxCoordMax = boxCoordMax / float( bricksInRefLevel[index] );
    return LODNode( internalNode, _volumeInfo.maximumBlockSize - _volumeInfo.overlap * 2u,
                    Boxf( boxCoordMin - _volumeInfo.worldSize * 0.5f,
                          boxCoordMax - _volumeInfo.worldSize * 0.5f ) );
}

/* ---- DataSource: livre/core/data/DataSource.cpp:38-112 ------------------------------------ */
DataSource::DataSource( const URI& uri )
    : _plugin( PluginFactory< DataSourcePlugin, const DataSourcePluginData& >::getInstance().create(
          DataSourcePluginData( uri ) ) )
{
}

DataSource::~DataSource() {}
const VolumeInformation& DataSource::getVolumeInfo() const { return _plugin->getVolumeInfo(); }
LODNode DataSource::getNode( const NodeId& nodeId ) const { return _plugin->getNode( nodeId ); }
void DataSource::update() { _plugin->update(); }

ConstMemoryUnitPtr DataSource::getData( const NodeId& nodeId )
{
    if( !nodeId.isValid() )
        return ConstMemoryUnitPtr();
    const LODNode lodNode = getNode( nodeId );
    if( !lodNode.isValid() )
        return ConstMemoryUnitPtr();
    return _plugin->getData( lodNode );
}
}

/*
 * data.h -- data model of the plugin surface: NodeId, RootNode, LODNode, VolumeInformation,
 * MemoryUnit, DataSource + DataSourcePlugin, PluginFactory.  Mirrors livre/core/data and
 * livre/core/util of the reference (citations per declaration, path:line from the reference
 * root); only what the raycast path and its callers use.
 */
#ifndef LIVRE_HIP_DATA_H
#define LIVRE_HIP_DATA_H

#include <functional>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "mathTypes.h"

namespace livre
{
typedef uint64_t Identifier; /* livre/core/types.h:87 */
typedef Identifier CacheId;  /* livre/core/types.h:88 */
const Identifier INVALID_CACHE_ID = Identifier( -1 );  /* types.h:186 */
const Identifier INVALID_NODE_ID = Identifier( -1 );   /* types.h:187 */
const uint32_t NODEID_LEVEL_BITS = 4;                  /* types.h:191 */
const uint32_t NODEID_BLOCK_BITS = 14;                 /* types.h:192 */
const uint32_t NODEID_TIMESTEP_BITS = 18;              /* mathTypes.h:82 */
const uint32_t INVALID_LEVEL = ( 1u << NODEID_LEVEL_BITS ) - 1u;       /* types.h:195 */
const uint32_t INVALID_TIMESTEP = ( 1u << NODEID_TIMESTEP_BITS ) - 1u; /* mathTypes.h:83 */
const Vector2ui FULL_FRAME_RANGE( 0, INVALID_TIMESTEP );               /* mathTypes.h:85 */

class NodeId;
typedef std::vector< NodeId > NodeIds;

/** 64-bit octree node identifier, livre/core/data/NodeId.h:35-130.
 *  Bit layout, LSB first: level:4 | x:14 | y:14 | z:14 | timeStep:18. */
class NodeId
{
public:
    NodeId() : _id( INVALID_NODE_ID ) {}
    explicit NodeId( const Identifier& identifier ) : _id( identifier ) {}
    NodeId( uint32_t level, const Vector3ui& position, uint32_t timeStep = 0 );

    uint32_t getLevel() const { return uint32_t( _id & 0xFu ); }
    uint32_t getTimeStep() const { return uint32_t( ( _id >> 46 ) & 0x3FFFFu ); }
    bool isRoot() const { return getLevel() == 0; }
    Vector3ui getPosition() const;
    NodeIds getParents() const;
    NodeId getParent() const;
    bool isParent( const NodeId& parentNodeId ) const;
    bool isChild( const NodeId& childNodeId ) const { return childNodeId.isParent( *this ); }
    bool isValid() const { return getLevel() != INVALID_LEVEL; }
    NodeIds getChildren() const;
    NodeId getRoot() const;
    NodeIds getSiblings() const;
    NodeIds getChildrenAtLevel( uint32_t level ) const;
    Range getRange() const;
    Identifier getId() const { return _id; }
    bool operator==( const NodeId& n ) const { return _id == n._id; }
    bool operator!=( const NodeId& n ) const { return _id != n._id; }
    bool operator<( const NodeId& n ) const { return _id < n._id; }

private:
    Identifier _id;
};

/** livre/core/data/NodeId.h:136-168 */
class RootNode
{
public:
    RootNode( uint32_t depth = 0, const Vector3ui& blockCount = Vector3ui( 0u ) )
        : _treeDepth( depth ), _blockCount( blockCount ) {}
    uint32_t getDepth() const { return _treeDepth; }
    Vector3ui getBlockSize( uint32_t level = 0 ) const { return _blockCount * ( 1u << level ); }

private:
    uint32_t _treeDepth;
    Vector3ui _blockCount;
};

/** livre/core/data/VolumeInformation.h:30-40 */
enum DataType { DT_FLOAT, DT_UINT8, DT_UINT16, DT_UINT32, DT_INT8, DT_INT16, DT_INT32, DT_UNDEFINED };

/** livre/core/data/VolumeInformation.h:43-112 */
struct VolumeInformation
{
    VolumeInformation();
    bool bigEndian;
    uint32_t compCount;
    DataType dataType;
    Vector3ui overlap;
    Vector3ui maximumBlockSize;
    Vector3ui voxels;
    Vector3f worldSize;
    Vector3f resolution;
    float worldSpacePerVoxel;
    float meterToDataUnitRatio;
    RootNode rootNode;
    size_t getBytesPerVoxel() const;
    Vector2ui frameRange;
    std::string description;
};

/** livre/core/data/DataSourcePlugin.cpp:83-109 */
bool fillRegularVolumeInfo( VolumeInformation& info );

/** livre/core/data/LODNode.h:35-124 */
class LODNode
{
public:
    LODNode() : _blockSize( 0u ) {}
    LODNode( const NodeId& nodeId, const Vector3ui& blockSize, const Boxf& worldBox );
    Vector3ui getAbsolutePosition() const { return _nodeId.getPosition(); }
    const Boxui& getVoxelBox() const { return _localVoxelBox; }
    const Boxf& getWorldBox() const { return _worldBox; }
    uint32_t getRefLevel() const { return _nodeId.getLevel(); }
    NodeId getNodeId() const { return _nodeId; }
    bool isValid() const { return _nodeId.isValid(); }
    const Vector3ui& getBlockSize() const { return _blockSize; }

private:
    NodeId _nodeId;
    Vector3ui _blockSize;
    Boxui _localVoxelBox;
    Boxf _worldBox;
};

/** livre/core/data/MemoryUnit.h:34-166 (const view + owning allocation) */
class MemoryUnit
{
public:
    virtual ~MemoryUnit() {}
    template < class T > const T* getData() const { return reinterpret_cast< const T* >( _getData() ); }
    virtual size_t getMemSize() const = 0;
    virtual size_t getAllocSize() const = 0;

protected:
    virtual const uint8_t* _getData() const = 0;
};
typedef std::shared_ptr< MemoryUnit > MemoryUnitPtr;
typedef std::shared_ptr< const MemoryUnit > ConstMemoryUnitPtr;

class ConstMemoryUnit : public MemoryUnit
{
public:
    ConstMemoryUnit( const uint8_t* ptr, size_t size ) : _ptr( ptr ), _size( size ) {}
    size_t getMemSize() const final { return _size; }
    size_t getAllocSize() const final { return 0; }

private:
    const uint8_t* _getData() const final { return _ptr; }
    const uint8_t* _ptr;
    size_t _size;
};

/** Owns its bytes.  The storage is NOT value-initialised (the data source overwrites all of it;
 * a zeroing pass over a 2.4 MiB brick costs as much as filling it) and, from 2 MiB up, is
 * 2 MiB-aligned and advised for transparent huge pages: a fresh brick then takes a couple of
 * page faults instead of six hundred -- first-touch faults were half of the upload time. */
class AllocMemoryUnit : public MemoryUnit
{
public:
    explicit AllocMemoryUnit( size_t size );
    ~AllocMemoryUnit();
    AllocMemoryUnit( const AllocMemoryUnit& ) = delete;
    AllocMemoryUnit& operator=( const AllocMemoryUnit& ) = delete;
    template < class T > T* getData() { return reinterpret_cast< T* >( _data ); }
    using MemoryUnit::getData;
    size_t getMemSize() const final { return _size; }
    size_t getAllocSize() const final { return _size; }

private:
    const uint8_t* _getData() const final { return _data; }
    uint8_t* _data;
    size_t _size;
};

/** minimal servus::URI: scheme://path?k=v&k2=v2#fragment */
class URI
{
public:
    explicit URI( const std::string& str );
    const std::string& getScheme() const { return _scheme; }
    const std::string& getPath() const { return _path; }
    const std::string& getFragment() const { return _fragment; }
    bool findQuery( const std::string& key, std::string& value ) const;
    const std::string& str() const { return _str; }

private:
    std::string _str, _scheme, _path, _fragment;
    std::vector< std::pair< std::string, std::string > > _query;
};

/** livre/core/data/DataSourcePluginData.h: what a data source plugin is constructed from */
class DataSourcePluginData
{
public:
    explicit DataSourcePluginData( const URI& uri ) : _uri( uri ) {}
    const URI& getURI() const { return _uri; }

private:
    URI _uri;
};

/** livre/core/data/DataSourcePlugin.h:38-110 */
class DataSourcePlugin
{
public:
    typedef DataSourcePlugin PluginT;
    DataSourcePlugin();
    virtual ~DataSourcePlugin() {}
    virtual MemoryUnitPtr getData( const LODNode& node ) = 0;
    LODNode getNode( const NodeId& nodeId ) const;
    const VolumeInformation& getVolumeInfo() const { return _volumeInfo; }
    virtual void update() {}
    /** DataSourcePlugin.cpp:55-81 */
    virtual LODNode internalNodeToLODNode( const NodeId& nodeId ) const;
    /** true (default) when internalNodeToLODNode is plain arithmetic: getNode then computes the
     *  node instead of going through the memoising map + read/write lock of
     *  DataSourcePlugin.cpp:29-48 (same result; the per-frame path calls getNode thousands of
     *  times).  A plugin with an expensive lookup (file-backed trees) returns false. */
    virtual bool nodeLookupIsCheap() const { return true; }

protected:
    VolumeInformation _volumeInfo;

private:
    mutable std::unordered_map< Identifier, LODNode > _lodNodeMap;
    mutable std::shared_timed_mutex _mutex;
};

/** Link-time plugin registry, livre/core/util/PluginFactory.h:54-126 reduced to
 *  "first registered plugin whose handles() is true wins; none -> std::runtime_error"
 *  (PluginFactory.ipp:41-49). */
template < class PluginT, class... Args > class PluginFactory
{
public:
    struct Holder
    {
        std::function< PluginT*( Args... ) > constructor;
        std::function< bool( Args... ) > handles;
    };
    static PluginFactory& getInstance()
    {
        static PluginFactory factory;
        return factory;
    }
    PluginT* create( Args... initData )
    {
        for( Holder& plugin : _plugins )
            if( plugin.handles( initData... ) )
                return plugin.constructor( initData... );
        throw std::runtime_error( "No plugin implementation available" );
    }
    void register_( const Holder& plugin ) { _plugins.push_back( plugin ); }
    void deregisterAll() { _plugins.clear(); }
    size_t size() const { return _plugins.size(); }

private:
    std::vector< Holder > _plugins;
};

/** livre/core/util/PluginRegisterer.h:36-56: a static instance registers Impl at load time */
template < typename Impl, class... Args > class PluginRegisterer
{
public:
    PluginRegisterer()
    {
        typename PluginFactory< typename Impl::PluginT, Args... >::Holder h;
        h.constructor = []( Args... args ) -> typename Impl::PluginT* { return new Impl( args... ); };
        h.handles = []( Args... args ) { return Impl::handles( args... ); };
        PluginFactory< typename Impl::PluginT, Args... >::getInstance().register_( h );
    }
};

/** livre/core/data/DataSource.h:38-110: facade creating the plugin that handles the URI */
class DataSource
{
public:
    explicit DataSource( const URI& uri );
    explicit DataSource( const std::string& uri ) : DataSource( URI( uri ) ) {}
    ~DataSource();
    const VolumeInformation& getVolumeInfo() const;
    LODNode getNode( const NodeId& nodeId ) const;
    ConstMemoryUnitPtr getData( const NodeId& nodeId ); /* DataSource.cpp:102-112 */
    void update();

private:
    std::unique_ptr< DataSourcePlugin > _plugin;
};
}
#endif

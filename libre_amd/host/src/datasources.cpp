/* datasources.cpp -- the DataSourcePlugin implementations the path is fed by:
 *   mem://  datasources/memory/MemoryDataSource.cpp (synthetic, constant value per brick)
 *   raw://  datasources/raw/RawDataSource.cpp       (mmap'd file, one brick = whole volume)
 *   hash:// build-defined seeded-noise volume ("Volume N" of SURVEY 8d), bricked like mem://
 * Registered at load time through static PluginRegisterer objects, as the reference does
 * (MemoryDataSource.cpp:46, RawDataSource.cpp:50). */
#include <thread>
#include <algorithm>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cerrno>
#include <cstring>
#include <ctime>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <vector>
#include <sstream>

#include "livre_hip/data.h"

namespace livre
{
namespace
{
std::vector< std::string > split( const std::string& s, char sep )
{
    std::vector< std::string > out;
    std::stringstream ss( s );
    std::string item;
    while( std::getline( ss, item, sep ) )
        out.push_back( item );
    return out;
}

bool endsWith( const std::string& s, const std::string& suffix )
{
    return s.size() >= suffix.size() && s.compare( s.size() - suffix.size(), suffix.size(), suffix ) == 0;
}

uint32_t toUint( const std::string& s )
{
    size_t pos = 0;
    const unsigned long v = std::stoul( s, &pos );
    if( pos != s.size() )
        throw std::runtime_error( "bad lexical cast: " + s );
    return uint32_t( v );
}
}

/* ---- mem:// -------------------------------------------------------------------------------- */
class MemoryDataSource : public DataSourcePlugin
{
public:
    explicit MemoryDataSource( const DataSourcePluginData& initData );
    MemoryUnitPtr getData( const LODNode& node ) final;
    static bool handles( const DataSourcePluginData& initData )
    {
        return initData.getURI().getScheme() == "mem"; /* MemoryDataSource.cpp:164-167 */
    }

private:
    float _sparsity;
};

/* MemoryDataSource.cpp:48-72 */
template < typename T >
static MemoryUnitPtr computeData( const LODNode& node, size_t dataSize, float sparsity, size_t nVoxels )
{
    const Identifier nodeId = node.getNodeId().getId();
    const uint8_t* id = reinterpret_cast< const uint8_t* >( &nodeId );
    const T value = T( ( id[0] ^ id[1] ^ id[2] ^ id[3] ) + 16 +
                       127 * std::sin( ( float( node.getNodeId().getTimeStep() ) + 1 ) / 200.f ) );
    std::shared_ptr< AllocMemoryUnit > memoryUnit( new AllocMemoryUnit( dataSize ) );
    T* dst = memoryUnit->getData< T >();
    if( sparsity < 1.f )
        for( size_t i = 0; i < nVoxels; ++i )
        {
            const int32_t random = rand() % 1000000 + 1;
            dst[i] = random < 1000000.0f * sparsity ? value : T( 0 );
        }
    else
        std::fill_n( dst, nVoxels, value );
    return memoryUnit;
}

/* MemoryDataSource.cpp:74-131 */
MemoryDataSource::MemoryDataSource( const DataSourcePluginData& initData ) : _sparsity( 1.0f )
{
    _volumeInfo.overlap = Vector3ui( 4 );
    const URI& uri = initData.getURI();
    const std::vector< std::string > parameters = split( uri.getFragment(), ',' );
    std::string v;
    try
    {
        if( uri.findQuery( "sparsity", v ) )
            _sparsity = std::stof( v );
        if( !uri.findQuery( "datatype", v ) || v == "uint8" ) _volumeInfo.dataType = DT_UINT8;
        else if( v == "uint16" ) _volumeInfo.dataType = DT_UINT16;
        else if( v == "uint32" ) _volumeInfo.dataType = DT_UINT32;
        else if( v == "int8" || v == "char" ) _volumeInfo.dataType = DT_INT8;
        else if( v == "int16" || v == "short" ) _volumeInfo.dataType = DT_INT16;
        else if( v == "int32" ) _volumeInfo.dataType = DT_INT32;
        else if( v == "float" ) _volumeInfo.dataType = DT_FLOAT;

        if( parameters.size() < 4 ) /* defaults */
        {
            _volumeInfo.voxels = Vector3ui( 4096 );
            _volumeInfo.maximumBlockSize = Vector3ui( 32 ) + _volumeInfo.overlap * 2u;
        }
        else
        {
            _volumeInfo.voxels[0] = toUint( parameters[0] );
            _volumeInfo.voxels[1] = toUint( parameters[1] );
            _volumeInfo.voxels[2] = toUint( parameters[2] );
            _volumeInfo.maximumBlockSize = Vector3ui( toUint( parameters[3] ) ) + _volumeInfo.overlap * 2u;
        }
    }
    catch( const std::exception& e )
    {
        throw std::runtime_error( e.what() );
    }
    _volumeInfo.frameRange = FULL_FRAME_RANGE;
    if( !fillRegularVolumeInfo( _volumeInfo ) )
        throw std::runtime_error( "Cannot setup the regular tree" );
}

/* MemoryDataSource.cpp:137-162 */
MemoryUnitPtr MemoryDataSource::getData( const LODNode& node )
{
    const Vector3ui blockSize = node.getBlockSize() + _volumeInfo.overlap * 2u;
    const size_t nVoxels = size_t( blockSize[0] ) * blockSize[1] * blockSize[2];
    const size_t dataSize = nVoxels * _volumeInfo.compCount * _volumeInfo.getBytesPerVoxel();
    switch( _volumeInfo.dataType )
    {
    case DT_UINT8: return computeData< uint8_t >( node, dataSize, _sparsity, nVoxels );
    case DT_UINT16: return computeData< uint16_t >( node, dataSize, _sparsity, nVoxels );
    case DT_UINT32: return computeData< uint32_t >( node, dataSize, _sparsity, nVoxels );
    case DT_INT8: return computeData< int8_t >( node, dataSize, _sparsity, nVoxels );
    case DT_INT16: return computeData< int16_t >( node, dataSize, _sparsity, nVoxels );
    case DT_INT32: return computeData< int32_t >( node, dataSize, _sparsity, nVoxels );
    case DT_FLOAT: return computeData< float >( node, dataSize, _sparsity, nVoxels );
    default: throw std::runtime_error( "Unimplemented data type." );
    }
}

/* ---- hash:// (build-defined): v(x,y,z) = lowbias32(x + vx*(y + vy*z) + seed) >> 24, 3-tap box
 * per axis with wrap-around, at the finest level; coarser levels sample every 2^k-th voxel.
 * Same URI fragment and tree as mem://. -------------------------------------------------------- */
class HashDataSource : public DataSourcePlugin
{
public:
    explicit HashDataSource( const DataSourcePluginData& initData )
    {
        _volumeInfo.overlap = Vector3ui( 4 );
        const std::vector< std::string > p = split( initData.getURI().getFragment(), ',' );
        if( p.size() < 4 )
            throw std::runtime_error( "hash://#x,y,z,block expected" );
        _volumeInfo.voxels = Vector3ui( toUint( p[0] ), toUint( p[1] ), toUint( p[2] ) );
        _volumeInfo.maximumBlockSize = Vector3ui( toUint( p[3] ) ) + _volumeInfo.overlap * 2u;
        _seed = p.size() > 4 ? toUint( p[4] ) : 0x5EEDu;
        _volumeInfo.dataType = DT_UINT8;
        _volumeInfo.frameRange = FULL_FRAME_RANGE;
        fillRegularVolumeInfo( _volumeInfo );
    }
    static bool handles( const DataSourcePluginData& d ) { return d.getURI().getScheme() == "hash"; }

    static uint32_t hash32( uint32_t h )
    {
        h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
        return h;
    }
    float raw( int64_t x, int64_t y, int64_t z ) const
    {
        const int64_t vx = _volumeInfo.voxels[0], vy = _volumeInfo.voxels[1], vz = _volumeInfo.voxels[2];
        x = ( x % vx + vx ) % vx; y = ( y % vy + vy ) % vy; z = ( z % vz + vz ) % vz;
        const uint64_t idx = uint64_t( x ) + uint64_t( vx ) * ( uint64_t( y ) + uint64_t( vy ) * uint64_t( z ) );
        return float( hash32( uint32_t( ( idx + _seed ) & 0xFFFFFFFFull ) ) >> 24 );
    }
    uint8_t voxel( int64_t x, int64_t y, int64_t z ) const
    {
        /* separable 3-tap box with wrap-around, filtered along z, then y, then x, each pass
         * ((prev + cur) + next) / 3 in float32 -- the order tests/orc.py:hash_volume uses */
        float fy[3];
        for( int dx = -1; dx <= 1; ++dx )
        {
            float fz[3];
            for( int dy = -1; dy <= 1; ++dy )
                fz[dy + 1] = ( ( raw( x + dx, y + dy, z - 1 ) + raw( x + dx, y + dy, z ) ) +
                               raw( x + dx, y + dy, z + 1 ) ) / 3.0f;
            fy[dx + 1] = ( ( fz[0] + fz[1] ) + fz[2] ) / 3.0f;
        }
        const float v = std::floor( ( ( fy[0] + fy[1] ) + fy[2] ) / 3.0f );
        return uint8_t( v < 0.f ? 0.f : ( v > 255.f ? 255.f : v ) );
    }
    MemoryUnitPtr getData( const LODNode& node ) final
    {
        const Vector3ui bs = node.getBlockSize() + _volumeInfo.overlap * 2u;
        std::shared_ptr< AllocMemoryUnit > mem( new AllocMemoryUnit( size_t( bs[0] ) * bs[1] * bs[2] ) );
        uint8_t* dst = mem->getData< uint8_t >();
        const uint32_t shift = _volumeInfo.rootNode.getDepth() - 1 - node.getRefLevel();
        const Vector3ui o = node.getVoxelBox().getMin();
        const Vector3ui vox = _volumeInfo.voxels;
        for( uint32_t z = 0; z < bs[2]; ++z )
            for( uint32_t y = 0; y < bs[1]; ++y )
                for( uint32_t x = 0; x < bs[0]; ++x )
                {
                    /* clamp at the volume border, like a bricking tool does */
                    int64_t gx = ( int64_t( o[0] ) + x - _volumeInfo.overlap[0] ) << shift;
                    int64_t gy = ( int64_t( o[1] ) + y - _volumeInfo.overlap[1] ) << shift;
                    int64_t gz = ( int64_t( o[2] ) + z - _volumeInfo.overlap[2] ) << shift;
                    gx = gx < 0 ? 0 : ( gx > int64_t( vox[0] ) - 1 ? int64_t( vox[0] ) - 1 : gx );
                    gy = gy < 0 ? 0 : ( gy > int64_t( vox[1] ) - 1 ? int64_t( vox[1] ) - 1 : gy );
                    gz = gz < 0 ? 0 : ( gz > int64_t( vox[2] ) - 1 ? int64_t( vox[2] ) - 1 : gz );
                    dst[( size_t( z ) * bs[1] + y ) * bs[0] + x] = voxel( gx, gy, gz );
                }
        return mem;
    }

private:
    uint32_t _seed;
};

/* ---- raw:// : datasources/raw/RawDataSource.cpp:56-129 ------------------------------------- */
class RawDataSource : public DataSourcePlugin
{
public:
    explicit RawDataSource( const DataSourcePluginData& initData )
        : _mmapPtr( nullptr ), _fd( -1 ), _size( 0 ), _dataOffset( 0 )
    {
        const URI& uri = initData.getURI();
        const std::string& path = uri.getPath();
        const bool isRaw = endsWith( path, ".raw" ) || endsWith( path, ".img" );
        const bool isNrrd = endsWith( path, ".nrrd" );
        if( !isRaw && !isNrrd )
            throw std::runtime_error( "Volume extension does not include raw or nrrd" );
        std::vector< std::string > p = split( uri.getFragment(), ',' );
        std::string dataFile = path;
        if( isNrrd )
        {
            /* RawDataSource.cpp:181-215 + nrrd/nrrd.hxx parseHeader: "field: value" lines up to the first
             * empty line; dimension must be 3, encoding raw; the voxels follow the header, or live in the
             * file named by "datafile" next to the header.  (The reference applies the header's length
             * to a detached data file as well, RawDataSource.cpp:189-197 + :126: bytes that are not a
             * header are skipped.  Here a detached file is read from its start, as the format says.) */
            std::map< std::string, std::string > fields;
            const size_t headerSize = parseNrrdHeader( path, fields );
            if( headerSize == 0 )
                throw std::runtime_error( "Cannot parse nrrd file" );
            if( fields.count( "encoding" ) && fields["encoding"] != "raw" )
                throw std::runtime_error( "NRRD encoding is not raw" );
            if( toUint( fields["dimension"] ) != 3u )
                throw std::runtime_error( "NRRD is not 3D data" );
            const std::vector< std::string > sizes = split( fields["sizes"], ' ' );
            if( sizes.size() < 3 )
                throw std::runtime_error( "NRRD sizes" );
            if( fields.count( "datafile" ) )
            {
                const size_t slash = path.find_last_of( '/' );
                dataFile = ( slash == std::string::npos ? std::string() : path.substr( 0, slash + 1 ) ) + fields["datafile"];
            }
            else
                _dataOffset = headerSize;
            _volumeInfo.bigEndian = fields.count( "endian" ) && fields["endian"] == "big";
            /* the URI fragment of a .nrrd may still carry the bricking extension: "#,,,,block" is awkward, so
             * a single number is taken as the block size */
            const std::string block = p.size() == 1 && !p[0].empty() ? p[0] : std::string();
            p = { sizes[0], sizes[1], sizes[2], nrrdTypeName( fields["type"] ) };
            if( !block.empty() )
                p.push_back( block );
        }
        _dataFile = dataFile;
        _fd = ::open( dataFile.c_str(), O_RDONLY );
        struct stat sb;
        if( _fd == -1 || ::fstat( _fd, &sb ) == -1 )
            throw std::runtime_error( "Cannot mmap file" );
        _size = size_t( sb.st_size );
        _mmapPtr = ::mmap( nullptr, _size, PROT_READ, MAP_PRIVATE, _fd, 0 );
        if( _mmapPtr == MAP_FAILED )
        {
            ::close( _fd );
            _mmapPtr = nullptr;
            throw std::runtime_error( "Cannot mmap file" );
        }
        if( p.size() < 4 )
            throw std::runtime_error( "Not enough parameters for the raw file" );
        _volumeInfo.voxels = Vector3ui( toUint( p[0] ), toUint( p[1] ), toUint( p[2] ) );
        setDataType( p[3] );
        _volumeInfo.frameRange = Vector2ui( 0u, 1u );
        _volumeInfo.compCount = 1;
        _volumeInfo.worldSpacePerVoxel = 1.0f / float( _volumeInfo.voxels.find_max() );
        _volumeInfo.worldSize = Vector3f( float( _volumeInfo.voxels[0] ), float( _volumeInfo.voxels[1] ),
                                          float( _volumeInfo.voxels[2] ) ) * _volumeInfo.worldSpacePerVoxel;
        _volumeInfo.overlap = Vector3ui( 0u );
        _volumeInfo.rootNode = RootNode( 1, Vector3ui( 1 ) ); /* one brick = whole volume, depth 1 */
        _volumeInfo.maximumBlockSize = _volumeInfo.voxels;
        if( _dataOffset > _size || size_t( _volumeInfo.voxels[0] ) * _volumeInfo.voxels[1] * _volumeInfo.voxels[2] *
                                       _volumeInfo.getBytesPerVoxel() > _size - _dataOffset )
            throw std::runtime_error( "raw file smaller than the declared volume" );
        /* EXTENSION (beyond the reference, whose raw source is the single brick above): a fifth
         * fragment parameter = block size turns the file into an out-of-core bricked volume with
         * the LOD tree, overlap and brick geometry of mem:// (DataSourcePlugin.cpp:55-109):
         * bricks are cut from the mapped file on demand (the page cache does the I/O), clamped at
         * the volume border; level l samples every 2^(depth-1-l)-th voxel. */
        _bricked = p.size() >= 5;
        if( _bricked )
        {
            const uint32_t block = toUint( p[4] );
            if( block == 0 )
                throw std::runtime_error( "raw://: block size 0" );
            _volumeInfo.overlap = Vector3ui( 4u );
            _volumeInfo.maximumBlockSize = Vector3ui( block ) + _volumeInfo.overlap * 2u;
            _volumeInfo.frameRange = FULL_FRAME_RANGE;
            /* fillRegularVolumeInfo derives world size and tree from voxels / block */
            if( !fillRegularVolumeInfo( _volumeInfo ) )
                throw std::runtime_error( "raw://: cannot build the LOD tree for this block size" );
#if defined( MADV_RANDOM )
            (void)::madvise( _mmapPtr, _size, MADV_RANDOM );
#endif
        }
    }
    ~RawDataSource()
    {
        if( _pyramidMap ) ::munmap( _pyramidMap, _pyramidMapSize );
        if( _mmapPtr ) ::munmap( _mmapPtr, _size );
        if( _fd != -1 ) ::close( _fd );
    }
    static bool handles( const DataSourcePluginData& d ) { return d.getURI().getScheme() == "raw"; }
    MemoryUnitPtr getData( const LODNode& node ) final
    {
        if( _bricked )
            return cutBrick( node );
        /* RawDataSource.cpp:123-129 reports blockSize.product() bytes whatever the voxel type
         * (quirk Q14); here the size includes bytes per voxel */
        const size_t dataSize = size_t( node.getBlockSize().product() ) * _volumeInfo.getBytesPerVoxel();
        return MemoryUnitPtr( new ConstMemoryUnit( static_cast< const uint8_t* >( _mmapPtr ) + _dataOffset, dataSize ) );
    }

private:
    /* nrrd/nrrd.hxx parseHeader: skip the magic line, then "field: value" / "key:=value" / "# comment" lines
     * up to the first empty line; returns the offset of the first byte after it (0: cannot read) */
    static size_t parseNrrdHeader( const std::string& path, std::map< std::string, std::string >& fields )
    {
        std::ifstream file( path.c_str(), std::ios::binary );
        if( !file.good() )
            return 0;
        std::string line;
        std::getline( file, line ); /* "NRRD000X" */
        size_t offset = 0;
        while( file.good() && std::getline( file, line ) )
        {
            if( file.tellg() != std::streampos( -1 ) )
                offset = size_t( file.tellg() );
            if( !line.empty() && line.back() == '\r' )
                line.pop_back();
            if( line.empty() ) /* beginning of the data chunk */
                break;
            if( line[0] == '#' )
                continue;
            const size_t colon = line.find( ':' );
            if( colon == std::string::npos || colon + 1 >= line.size() )
                return 0;
            if( line[colon + 1] == ' ' )
                fields[line.substr( 0, colon )] = line.substr( colon + 2 );
            else if( line[colon + 1] != '=' )
                return 0;
        }
        return offset;
    }
    /* nrrd.hxx maps every NRRD type name to its C type; RawDataSource::setDataType takes those */
    static std::string nrrdTypeName( const std::string& t )
    {
        static const char* names[][2] = {
            { "int8_t", "char" }, { "int8", "char" }, { "signed char", "char" }, { "char", "char" },
            { "uchar", "unsigned char" }, { "uint8_t", "unsigned char" }, { "uint8", "unsigned char" },
            { "unsigned char", "unsigned char" },
            { "int16_t", "short" }, { "int16", "short" }, { "signed short int", "short" }, { "short int", "short" },
            { "short", "short" },
            { "ushort", "unsigned short" }, { "uint16_t", "unsigned short" }, { "uint16", "unsigned short" },
            { "unsigned short int", "unsigned short" }, { "unsigned short", "unsigned short" },
            { "int32_t", "int" }, { "int32", "int" }, { "signed int", "int" }, { "int", "int" },
            { "uint32_t", "unsigned int" }, { "uint32", "unsigned int" }, { "uint", "unsigned int" },
            { "unsigned int", "unsigned int" }, { "float", "float" } };
        for( const auto& n : names )
            if( t == n[0] )
                return n[1];
        throw std::runtime_error( "Cannot parse nrrd file" ); /* parseHeader returns 0 for an unknown type */
    }

    /* brick + overlap of one LOD node, cut out of the mapped volume */
    MemoryUnitPtr cutBrick( const LODNode& node ) const
    {
        const size_t bpv = _volumeInfo.getBytesPerVoxel();
        const Vector3ui bs = node.getBlockSize() + _volumeInfo.overlap * 2u;
        std::shared_ptr< AllocMemoryUnit > mem( new AllocMemoryUnit( size_t( bs[0] ) * bs[1] * bs[2] * bpv ) );
        uint8_t* dst = mem->getData< uint8_t >();
        const uint8_t* src = static_cast< const uint8_t* >( _mmapPtr ) + _dataOffset;
        uint32_t shift = _volumeInfo.rootNode.getDepth() - 1 - node.getRefLevel();
        const Vector3ui o = node.getVoxelBox().getMin();
        int64_t vx = _volumeInfo.voxels[0], vy = _volumeInfo.voxels[1], vz = _volumeInfo.voxels[2];
        /* coarse levels come from the in-memory pyramid (every 2^k-th voxel of the file, built
         * level by level on first use: strided reads of the mapped file cost ~12 ms per coarse
         * brick, a cut from the decimated copy as little as a finest-level one) */
        if( shift > 0 )
        {
            const PyramidLevel& lvl = pyramidLevel( shift );
            src = lvl.data;
            vx = lvl.dim[0];
            vy = lvl.dim[1];
            vz = lvl.dim[2];
            shift = 0;
        }
        /* source x of every brick column, clamped at the volume border */
        std::vector< int64_t > sx( bs[0] );
        for( uint32_t x = 0; x < bs[0]; ++x )
        {
            const int64_t g = ( int64_t( o[0] ) + x - _volumeInfo.overlap[0] ) << shift;
            sx[x] = g < 0 ? 0 : ( g > vx - 1 ? vx - 1 : g );
        }
        /* the interior run of a finest-level row is contiguous in the file */
        uint32_t runBegin = 0, runEnd = 0;
        if( shift == 0 )
        {
            while( runBegin < bs[0] && ( int64_t( o[0] ) + runBegin ) < int64_t( _volumeInfo.overlap[0] ) )
                ++runBegin;
            runEnd = runBegin;
            while( runEnd < bs[0] && sx[runEnd] == sx[runBegin] + int64_t( runEnd - runBegin ) )
                ++runEnd;
        }
        for( uint32_t z = 0; z < bs[2]; ++z )
        {
            int64_t gz = ( int64_t( o[2] ) + z - _volumeInfo.overlap[2] ) << shift;
            gz = gz < 0 ? 0 : ( gz > vz - 1 ? vz - 1 : gz );
            for( uint32_t y = 0; y < bs[1]; ++y )
            {
                int64_t gy = ( int64_t( o[1] ) + y - _volumeInfo.overlap[1] ) << shift;
                gy = gy < 0 ? 0 : ( gy > vy - 1 ? vy - 1 : gy );
                const uint8_t* row = src + size_t( ( gz * vy + gy ) * vx ) * bpv;
                uint8_t* out = dst + ( size_t( z ) * bs[1] + y ) * bs[0] * bpv;
                for( uint32_t x = 0; x < runBegin; ++x )
                    std::memcpy( out + x * bpv, row + size_t( sx[x] ) * bpv, bpv );
                if( runEnd > runBegin )
                    std::memcpy( out + runBegin * bpv, row + size_t( sx[runBegin] ) * bpv,
                                 size_t( runEnd - runBegin ) * bpv );
                for( uint32_t x = runEnd; x < bs[0]; ++x )
                    std::memcpy( out + x * bpv, row + size_t( sx[x] ) * bpv, bpv );
            }
        }
        return mem;
    }

private:
    void setDataType( const std::string& t )
    {
        if( t == "char" || t == "int8" ) _volumeInfo.dataType = DT_INT8;
        else if( t == "unsigned char" || t == "uint8" ) _volumeInfo.dataType = DT_UINT8;
        else if( t == "short" || t == "int16" ) _volumeInfo.dataType = DT_INT16;
        else if( t == "unsigned short" || t == "uint16" ) _volumeInfo.dataType = DT_UINT16;
        else if( t == "int" || t == "int32" ) _volumeInfo.dataType = DT_INT32;
        else if( t == "unsigned int" || t == "uint32" ) _volumeInfo.dataType = DT_UINT32;
        else if( t == "float" ) _volumeInfo.dataType = DT_FLOAT;
        else throw std::runtime_error( "Not supported data format" );
    }
    /* level k of the pyramid: voxel (x,y,z) = file voxel (x<<k, y<<k, z<<k) */
    static constexpr uint32_t kMaxPyramidLevels = 32u;
    struct PyramidLevel
    {
        std::unique_ptr< uint8_t[] > owned; /* built in this process ... */
        const uint8_t* data = nullptr;      /* ... or mapped from the pyramid file next to the volume */
        int64_t dim[3] = { 0, 0, 0 };
    };
    const PyramidLevel& pyramidLevel( uint32_t k ) const
    {
        std::lock_guard< std::mutex > lock( _pyramidMutex );
        if( !_pyramidFileTried )
        {
            _pyramidFileTried = true;
            mapPyramidFile(); /* levels a previous run left on disk */
        }
        /* whatever is missing is built up to the coarsest level of the tree (each level is an eighth of the one
         * below): the file then holds the whole pyramid */
        const uint32_t depth = _volumeInfo.rootNode.getDepth();
        const uint32_t top = std::max( k, depth > 0u ? depth - 1u : 0u );
        /* references to levels are handed out and used after the lock is gone: the vector never reallocates */
        if( top >= kMaxPyramidLevels )
            throw std::runtime_error( "raw://: LOD pyramid of more than 32 levels" );
        if( _pyramid.capacity() < kMaxPyramidLevels )
            _pyramid.reserve( kMaxPyramidLevels );
        if( _pyramid.size() <= top )
            _pyramid.resize( top + 1 );
        bool built = false;
        for( uint32_t l = 1; l <= top; ++l )
        {
            if( _pyramid[l].data )
                continue;
            built = true;
            const size_t bpv = _volumeInfo.getBytesPerVoxel();
            const uint8_t* prev = l == 1 ? static_cast< const uint8_t* >( _mmapPtr ) + _dataOffset : _pyramid[l - 1].data;
            int64_t pd[3];
            for( int a = 0; a < 3; ++a )
                pd[a] = l == 1 ? int64_t( _volumeInfo.voxels[a] ) : _pyramid[l - 1].dim[a];
            PyramidLevel& lv = _pyramid[l];
            for( int a = 0; a < 3; ++a )
                lv.dim[a] = ( pd[a] + 1 ) / 2; /* voxels 0, 2, 4, ... of the level below */
            lv.owned.reset( new uint8_t[size_t( lv.dim[0] ) * lv.dim[1] * lv.dim[2] * bpv] );
            lv.data = lv.owned.get();
            /* z-slabs on all host cores: level 1 of a 2048^3 uint16 file is 1e9 voxels picked out of
             * 17 GB (3.2 s on one core) */
            uint8_t* const dst = lv.owned.get();
            const int64_t d0 = lv.dim[0], d1 = lv.dim[1], d2 = lv.dim[2], p0 = pd[0], p1 = pd[1];
            auto slab = [=]( int64_t z0, int64_t z1 ) {
                for( int64_t z = z0; z < z1; ++z )
                    for( int64_t y = 0; y < d1; ++y )
                    {
                        const uint8_t* row = prev + size_t( ( ( 2 * z ) * p1 + 2 * y ) * p0 ) * bpv;
                        uint8_t* out = dst + size_t( ( z * d1 + y ) * d0 ) * bpv;
                        if( bpv == 1 )
                            for( int64_t x = 0; x < d0; ++x )
                                out[x] = row[2 * x];
                        else if( bpv == 2 )
                            for( int64_t x = 0; x < d0; ++x )
                                std::memcpy( out + 2 * x, row + 4 * x, 2 );
                        else
                            for( int64_t x = 0; x < d0; ++x )
                                std::memcpy( out + x * bpv, row + size_t( 2 * x ) * bpv, bpv );
                    }
            };
            const int64_t nThreads = std::max< int64_t >(
                1, std::min< int64_t >( { int64_t( std::thread::hardware_concurrency() ), int64_t( 16 ), d2 / 8 } ) );
            if( nThreads == 1 )
                slab( 0, d2 );
            else
            {
                std::vector< std::thread > workers;
                for( int64_t t = 0; t < nThreads; ++t )
                    workers.emplace_back( slab, d2 * t / nThreads, d2 * ( t + 1 ) / nThreads );
                for( std::thread& w : workers )
                    w.join();
            }
        }
        /* the whole pyramid is in memory for the first time: leave it on disk for the next run */
        if( built && !_pyramidMap )
            writePyramidFile();
        return _pyramid[k];
    }

    /* ---- the pyramid on disk (round 3; opt-in since round 4) ---------------------------------------------------
     * OFF by default: a renderer does not leave files in the user's data directory unasked (round-3 advisor).
     * LIVRE_HIP_PYRAMID_DIR=<dir> keeps <dir>/<file name>.lvpyr; LIVRE_HIP_PYRAMID=1 keeps <volume file>.lvpyr next to
     * the volume.  Header { "LVPYR001", size and mtime of the volume file, voxels, bytes per voxel, data offset,
     * levels }, then per level { dim[3], offset }, then the levels, 4096-aligned.  Written once by the first process
     * that has built every level -- through <path>.writing, created with O_EXCL, so that of N ranks starting together
     * one writes and the others go on with what they built (a .writing file older than ten minutes is a crashed
     * writer's and is replaced) -- then renamed; best effort: a read-only directory just means the next run builds
     * again.  Mapped by later runs after the header has been checked against the volume file as it is now.  The
     * 2048^3 uint16 volume of BASELINE C3 spends 0.26 s of every start-up on the levels otherwise. */
    struct PyramidFileHeader
    {
        char magic[8];
        uint64_t sourceSize, sourceMtimeNs, dataOffset;
        uint32_t voxels[3], bytesPerVoxel, levels, reserved;
    };
    struct PyramidFileLevel
    {
        uint64_t dim[3], offset;
    };
    std::string pyramidPath() const
    {
        const char* on = ::getenv( "LIVRE_HIP_PYRAMID" );
        if( on && on[0] == '0' )
            return std::string();
        const char* dir = ::getenv( "LIVRE_HIP_PYRAMID_DIR" );
        if( dir && dir[0] )
        {
            const size_t slash = _dataFile.find_last_of( '/' );
            return std::string( dir ) + "/" + ( slash == std::string::npos ? _dataFile : _dataFile.substr( slash + 1 ) ) + ".lvpyr";
        }
        if( on && on[0] == '1' )
            return _dataFile + ".lvpyr";
        return std::string(); /* not asked for */
    }
    bool fillHeader( PyramidFileHeader& h ) const
    {
        struct stat st;
        if( ::fstat( _fd, &st ) != 0 )
            return false;
        std::memset( &h, 0, sizeof( h ) );
        std::memcpy( h.magic, "LVPYR001", 8 );
        h.sourceSize = uint64_t( st.st_size );
        h.sourceMtimeNs = uint64_t( st.st_mtim.tv_sec ) * 1000000000ull + uint64_t( st.st_mtim.tv_nsec );
        h.dataOffset = _dataOffset;
        for( int a = 0; a < 3; ++a )
            h.voxels[a] = _volumeInfo.voxels[a];
        h.bytesPerVoxel = uint32_t( _volumeInfo.getBytesPerVoxel() );
        h.levels = _volumeInfo.rootNode.getDepth() - 1u;
        return true;
    }
    void mapPyramidFile() const
    {
        const std::string path = pyramidPath();
        PyramidFileHeader want;
        if( path.empty() || !fillHeader( want ) || want.levels == 0 )
            return;
        const int fd = ::open( path.c_str(), O_RDONLY );
        if( fd == -1 )
            return;
        struct stat st;
        if( ::fstat( fd, &st ) != 0 || size_t( st.st_size ) < sizeof( PyramidFileHeader ) )
        {
            ::close( fd );
            return;
        }
        void* map = ::mmap( nullptr, size_t( st.st_size ), PROT_READ, MAP_PRIVATE, fd, 0 );
        ::close( fd );
        if( map == MAP_FAILED )
            return;
        const uint8_t* base = static_cast< const uint8_t* >( map );
        bool ok = std::memcmp( base, &want, sizeof( want ) ) == 0 &&
                  size_t( st.st_size ) >= sizeof( want ) + size_t( want.levels ) * sizeof( PyramidFileLevel );
        std::vector< PyramidLevel > levels( want.levels + 1u );
        if( ok )
        {
            const PyramidFileLevel* tl = reinterpret_cast< const PyramidFileLevel* >( base + sizeof( want ) );
            int64_t pd[3] = { int64_t( want.voxels[0] ), int64_t( want.voxels[1] ), int64_t( want.voxels[2] ) };
            for( uint32_t l = 1; ok && l <= want.levels; ++l )
            {
                uint64_t bytes = want.bytesPerVoxel;
                for( int a = 0; a < 3; ++a )
                {
                    pd[a] = ( pd[a] + 1 ) / 2;
                    ok = ok && tl[l - 1].dim[a] == uint64_t( pd[a] );
                    bytes *= uint64_t( pd[a] );
                    levels[l].dim[a] = pd[a];
                }
                ok = ok && tl[l - 1].offset <= uint64_t( st.st_size ) && bytes <= uint64_t( st.st_size ) - tl[l - 1].offset;
                levels[l].data = base + tl[l - 1].offset;
            }
        }
        if( !ok ) /* another volume, an older file, a truncated write: build again (and overwrite) */
        {
            ::munmap( map, size_t( st.st_size ) );
            return;
        }
        _pyramidMap = map;
        _pyramidMapSize = size_t( st.st_size );
        _pyramid = std::move( levels );
    }
    void writePyramidFile() const
    {
        const std::string path = pyramidPath();
        PyramidFileHeader h;
        if( path.empty() || !fillHeader( h ) || h.levels == 0 || _pyramid.size() <= h.levels )
            return;
        const std::string tmp = path + ".writing";
        int wfd = ::open( tmp.c_str(), O_CREAT | O_EXCL | O_WRONLY, 0644 );
        if( wfd == -1 && errno == EEXIST )
        {
            struct stat ts;
            if( ::stat( tmp.c_str(), &ts ) == 0 && ::time( nullptr ) - ts.st_mtime > 600 && ::unlink( tmp.c_str() ) == 0 )
                wfd = ::open( tmp.c_str(), O_CREAT | O_EXCL | O_WRONLY, 0644 );
        }
        if( wfd == -1 )
            return; /* another process is writing it, or the directory is not ours to write */
        FILE* f = ::fdopen( wfd, "wb" );
        if( !f )
        {
            ::close( wfd );
            std::remove( tmp.c_str() );
            return;
        }
        std::vector< PyramidFileLevel > tl( h.levels );
        uint64_t at = ( sizeof( h ) + h.levels * sizeof( PyramidFileLevel ) + 4095u ) & ~uint64_t( 4095 );
        for( uint32_t l = 1; l <= h.levels; ++l )
        {
            uint64_t bytes = h.bytesPerVoxel;
            for( int a = 0; a < 3; ++a )
            {
                tl[l - 1].dim[a] = uint64_t( _pyramid[l].dim[a] );
                bytes *= uint64_t( _pyramid[l].dim[a] );
            }
            tl[l - 1].offset = at;
            at = ( at + bytes + 4095u ) & ~uint64_t( 4095 );
        }
        bool ok = std::fwrite( &h, sizeof( h ), 1, f ) == 1 &&
                  std::fwrite( tl.data(), sizeof( PyramidFileLevel ), tl.size(), f ) == tl.size();
        for( uint32_t l = 1; ok && l <= h.levels; ++l )
        {
            uint64_t bytes = h.bytesPerVoxel;
            for( int a = 0; a < 3; ++a )
                bytes *= uint64_t( _pyramid[l].dim[a] );
            ok = std::fseek( f, long( tl[l - 1].offset ), SEEK_SET ) == 0 &&
                 std::fwrite( _pyramid[l].data, 1, size_t( bytes ), f ) == size_t( bytes );
        }
        ok = ( std::fclose( f ) == 0 ) && ok;
        if( !ok || std::rename( tmp.c_str(), path.c_str() ) != 0 )
            std::remove( tmp.c_str() );
    }

    void* _mmapPtr;
    int _fd;
    size_t _size;
    size_t _dataOffset; /* .nrrd with the voxels behind the header: where they start */
    bool _bricked = false;
    mutable std::mutex _pyramidMutex;
    mutable std::vector< PyramidLevel > _pyramid;
    mutable bool _pyramidFileTried = false;
    mutable void* _pyramidMap = nullptr;
    mutable size_t _pyramidMapSize = 0;
    std::string _dataFile; /* the file the voxels are in (the .nrrd's data file when detached) */
};

namespace
{
PluginRegisterer< MemoryDataSource, const DataSourcePluginData& > memRegisterer;
PluginRegisterer< RawDataSource, const DataSourcePluginData& > rawRegisterer;
PluginRegisterer< HashDataSource, const DataSourcePluginData& > hashRegisterer;
}
}

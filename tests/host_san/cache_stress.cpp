/* ThreadSanitizer stress of the mirrored livre::Cache (Cache.ipp semantics) with the DataObject
 * cache over a mem:// data source: concurrent load / get / unload with an LRU budget far below
 * the working set.  TEST INFRASTRUCTURE ONLY (tests/test_host.py::test_cache_under_thread_sanitizer). */
#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

#include "livre_hip/cache.h"
#include "livre_hip/data.h"

using namespace livre;

int main()
{
    DataSource source{ std::string( "mem://#128,128,128,16" ) }; /* 512 leaf bricks of 24^3 */
    const size_t brick = 24 * 24 * 24;
    DataCache cache( "stress", 40 * brick ); /* room for ~40 of them */
    std::vector< NodeId > ids;
    for( uint32_t z = 0; z < 8; ++z )
        for( uint32_t y = 0; y < 8; ++y )
            for( uint32_t x = 0; x < 8; ++x )
                ids.push_back( NodeId( 3, Vector3ui( x, y, z ), 0 ) );
    std::atomic< size_t > loads( 0 ), hits( 0 ), bad( 0 );
    std::vector< std::thread > threads;
    for( int t = 0; t < 8; ++t )
        threads.emplace_back( [&, t] {
            uint32_t s = 12345u + 77u * uint32_t( t );
            for( int i = 0; i < 4000; ++i )
            {
                s = s * 1664525u + 1013904223u;
                const NodeId& id = ids[( s >> 8 ) % ids.size()];
                if( ( s & 3u ) == 0 )
                {
                    if( cache.get( id.getId() ) )
                        ++hits;
                }
                else if( ( s & 3u ) == 1 )
                    cache.unload( id.getId() );
                else
                {
                    const auto obj = cache.load( id.getId(), source );
                    if( obj )
                    {
                        ++loads;
                        const auto data = std::static_pointer_cast< const DataObject >( obj );
                        /* the brick of node (x,y,z) is constant: first and last byte agree */
                        const uint8_t* p = static_cast< const uint8_t* >( data->getDataPtr() );
                        if( data->getMemSize() != brick || p[0] != p[brick - 1] )
                            ++bad;
                    }
                }
            }
        } );
    for( auto& th : threads )
        th.join();
    const CacheStatistics& st = cache.getStatistics();
    std::printf( "loads %zu hits %zu bad %zu used %zu max %zu count %zu\n", loads.load(), hits.load(), bad.load(),
                 st.getUsedMemory(), st.getMaximumMemory(), cache.getCount() );
    if( bad != 0 || st.getUsedMemory() > st.getMaximumMemory() + 8 * brick )
        return 1;
    std::printf( "DONE\n" );
    return 0;
}

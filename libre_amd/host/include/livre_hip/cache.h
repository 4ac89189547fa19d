/*
 * cache.h -- CacheObject, CacheStatistics, thread-safe LRU Cache<T>, DataObject.
 * Mirrors livre/core/cache/{CacheObject.h,Cache.h,Cache.ipp,CacheStatistics.h} and
 * livre/lib/cache/DataObject.h of the reference.
 */
#ifndef LIVRE_HIP_CACHE_H
#define LIVRE_HIP_CACHE_H

#include <atomic>
#include <deque>
#include <list>
#include <exception>
#include <sstream>

#include "data.h"

namespace livre
{
/** livre/core/cache/CacheObject.h:32-44 */
class CacheLoadException : public std::exception
{
public:
    CacheLoadException( const Identifier& id, const std::string& message )
    {
        std::stringstream s;
        s << "Id: " << id << " " << message;
        _message = s.str();
    }
    const char* what() const throw() { return _message.c_str(); }

private:
    std::string _message;
};

/** livre/core/cache/CacheObject.h:49-74 */
class CacheObject
{
public:
    virtual ~CacheObject() {}
    virtual size_t getSize() const = 0;
    CacheId getId() const { return _cacheId; }
    virtual bool operator==( const CacheObject& o ) const { return _cacheId == o._cacheId; }

protected:
    explicit CacheObject( const CacheId& cacheId = INVALID_CACHE_ID ) : _cacheId( cacheId ) {}

private:
    CacheId _cacheId;
};
typedef std::shared_ptr< CacheObject > CacheObjectPtr;
typedef std::shared_ptr< const CacheObject > ConstCacheObjectPtr;
typedef std::vector< ConstCacheObjectPtr > ConstCacheObjects;
typedef std::vector< CacheId > CacheIds;
typedef std::unordered_map< CacheId, ConstCacheObjectPtr > ConstCacheMap;

/** livre/core/cache/CacheStatistics.h:34-109 */
class CacheStatistics
{
public:
    CacheStatistics( const std::string& name, size_t maxMemBytes )
        : _name( name ), _usedMemBytes( 0 ), _maxMemBytes( maxMemBytes ), _objCount( 0 ),
          _cacheHit( 0 ), _cacheMiss( 0 ) {}
    size_t getBlockCount() const { return _objCount; }
    size_t getUsedMemory() const { return _usedMemBytes; }
    size_t getMaximumMemory() const { return _maxMemBytes; }
    std::string getName() const { return _name; }
    size_t getHits() const { return _cacheHit; }
    size_t getMisses() const { return _cacheMiss; }
    void notifyMiss() { ++_cacheMiss; }
    void notifyHit() { ++_cacheHit; }
    void notifyLoaded( const CacheObject& o ) { ++_objCount; _usedMemBytes += o.getSize(); }
    void notifyUnloaded( const CacheObject& o ) { --_objCount; _usedMemBytes -= o.getSize(); }
    void clear() { _usedMemBytes = 0; _objCount = 0; _cacheHit = 0; _cacheMiss = 0; }

private:
    std::string _name;
    /* written under the cache's write lock, read by monitoring code without it */
    std::atomic< size_t > _usedMemBytes;
    const size_t _maxMemBytes;
    std::atomic< size_t > _objCount, _cacheHit, _cacheMiss;
};

/** livre/core/cache/Cache.h:38-101 + Cache.ipp.  LRU by load order; an object is evicted only
 *  when nobody outside the cache holds it (Cache.ipp:207-220: use_count() <= 2 counting the
 *  map's reference and the local copy). */
template < class CacheObjectT > class Cache
{
public:
    typedef std::shared_ptr< const CacheObjectT > ObjectPtr;

    Cache( const std::string& name, size_t maxMemBytes )
        : _maxMemBytes( maxMemBytes ), _statistics( name, maxMemBytes ) {}

    ObjectPtr get( const CacheId& cacheId ) const
    {
        std::shared_lock< std::shared_timed_mutex > lock( _mutex );
        const auto it = _cacheMap.find( cacheId );
        return it == _cacheMap.end() ? ObjectPtr() : it->second->obj;
    }

    /** get() for a list of ids under ONE read lock (a frame asks for hundreds of bricks): out[k]
     *  belongs to ids[k], empty where the object is not in the cache */
    template < class IdRange > void getMany( const IdRange& ids, std::vector< ObjectPtr >& out ) const
    {
        out.clear();
        out.reserve( ids.size() );
        std::shared_lock< std::shared_timed_mutex > lock( _mutex );
        for( const auto& id : ids )
        {
            const auto it = _cacheMap.find( CacheId( id ) );
            out.push_back( it == _cacheMap.end() ? ObjectPtr() : it->second->obj );
        }
    }

    /** Cache.ipp:146-195: create-if-absent; constructor runs outside the map lock, under the
     *  entry's own lock, so concurrent loads of the same id construct once and of different
     *  ids run in parallel; a CacheLoadException yields an empty pointer (Cache.ipp:98-115). */
    template < class... Args > ObjectPtr load( const CacheId& cacheId, Args&&... args )
    {
        if( cacheId == INVALID_CACHE_ID )
            return ObjectPtr();
        {   /* hit: read lock only (Cache.ipp:149-155 also takes the read lock first) */
            std::shared_lock< std::shared_timed_mutex > lock( _mutex );
            const auto it = _cacheMap.find( cacheId );
            if( it != _cacheMap.end() && it->second->obj )
            {
                _hits.fetch_add( 1, std::memory_order_relaxed );
                return it->second->obj;
            }
        }
        std::shared_ptr< Entry > entry;
        {
            std::unique_lock< std::shared_timed_mutex > lock( _mutex );
            auto it = _cacheMap.find( cacheId );
            if( it == _cacheMap.end() )
                it = _cacheMap.emplace( cacheId, std::make_shared< Entry >() ).first;
            entry = it->second;
            if( entry->obj )
            {
                _hits.fetch_add( 1, std::memory_order_relaxed );
                return entry->obj;
            }
            applyPolicy();
        }
        /* Entry::obj is written with BOTH the entry's mutex and the map's write lock held, so it
         * may be read under either: here under the entry's mutex, in get()/the hit path under the
         * map's read lock, in unloadLocked() under its write lock (ThreadSanitizer-checked:
         * tests/host_san/cache_stress.cpp). */
        std::lock_guard< std::mutex > elock( entry->mutex );
        if( entry->obj ) /* another loader of the same id was first */
            return entry->obj;
        ObjectPtr created;
        try
        {
            created.reset( new CacheObjectT( cacheId, args... ) );
        }
        catch( const CacheLoadException& )
        {
        }
        std::unique_lock< std::shared_timed_mutex > lock( _mutex );
        const auto it = _cacheMap.find( cacheId );
        if( !created )
        {
            if( it != _cacheMap.end() && it->second == entry && !entry->obj )
                _cacheMap.erase( it );
            return ObjectPtr();
        }
        entry->obj = created;
        if( it == _cacheMap.end() )
            _cacheMap.emplace( cacheId, entry ); /* the empty entry was unloaded meanwhile */
        else if( it->second != entry )
            return it->second->obj ? it->second->obj : created; /* replaced meanwhile: keep theirs */
        _statistics.notifyMiss();
        _statistics.notifyLoaded( *created );
        lruInsert( cacheId );
        applyPolicy();
        return created;
    }

    /** Cache.ipp:222-240 */
    bool unload( const CacheId& cacheId )
    {
        std::unique_lock< std::shared_timed_mutex > lock( _mutex );
        return unloadLocked( cacheId );
    }

    size_t getCount() const
    {
        std::shared_lock< std::shared_timed_mutex > lock( _mutex );
        return _cacheMap.size();
    }
    const CacheStatistics& getStatistics() const { return _statistics; }
    size_t getHits() const { return _hits.load(); }

    void purge()
    {
        std::unique_lock< std::shared_timed_mutex > lock( _mutex );
        _statistics.clear();
        _lru.clear();
        _lruPos.clear();
        _cacheMap.clear();
    }
    void purge( const CacheId& cacheId )
    {
        std::unique_lock< std::shared_timed_mutex > lock( _mutex );
        _cacheMap.erase( cacheId );
    }

private:
    struct Entry
    {
        std::mutex mutex;
        ObjectPtr obj;
    };

    /* least recently loaded first (Cache.ipp:146-166 keeps a deque and scans it); list + index: O(1) */
    void lruInsert( const CacheId& id )
    {
        lruRemove( id );
        _lruPos[id] = _lru.insert( _lru.end(), id );
    }
    void lruRemove( const CacheId& id )
    {
        const auto it = _lruPos.find( id );
        if( it == _lruPos.end() )
            return;
        _lru.erase( it->second );
        _lruPos.erase( it );
    }
    bool unloadLocked( const CacheId& cacheId )
    {
        const auto it = _cacheMap.find( cacheId );
        if( it == _cacheMap.end() )
            return false;
        const ObjectPtr obj = it->second->obj; /* +1 reference, as in Cache.ipp:212 */
        if( !obj || obj.use_count() > 2 )      /* still loading or referenced outside */
            return false;
        lruRemove( cacheId );
        _statistics.notifyUnloaded( *obj );
        _cacheMap.erase( it );
        return true;
    }
    /** Cache.ipp:132-144: when used >= max, unload in LRU order until used < max */
    void applyPolicy()
    {
        if( _cacheMap.empty() || _statistics.getUsedMemory() < _maxMemBytes )
            return;
        for( auto it = _lru.begin(); it != _lru.end(); )
        {
            const CacheId id = *it;
            ++it; /* unloadLocked erases the element it is given, nothing else */
            unloadLocked( id );
            if( _statistics.getUsedMemory() < _maxMemBytes )
                return;
        }
    }

    const size_t _maxMemBytes;
    CacheStatistics _statistics;
    std::atomic< size_t > _hits{ 0 };
    mutable std::shared_timed_mutex _mutex;
    std::unordered_map< CacheId, std::shared_ptr< Entry > > _cacheMap;
    std::list< CacheId > _lru;
    std::unordered_map< CacheId, typename std::list< CacheId >::iterator > _lruPos;
};

/** livre/lib/cache/DataObject.h:31-60: a brick in CPU memory */
class DataObject : public CacheObject
{
public:
    DataObject( const CacheId& cacheId, DataSource& dataSource );
    size_t getSize() const final { return _data->getAllocSize(); }
    const void* getDataPtr() const { return _data->getData< void >(); }
    size_t getMemSize() const { return _data->getMemSize(); }

private:
    ConstMemoryUnitPtr _data;
};
typedef std::shared_ptr< const DataObject > ConstDataObjectPtr;
typedef Cache< DataObject > DataCache;
}
#endif

"""Pure-Python restatement of the reference block allocator (small cases only).

TEST INFRASTRUCTURE ONLY (see oracle/nvllm_oracle.c header).

Follows /root/reference/src/core/block_manager.rs:40-383 and
src/core/block.rs:72-156 statement by statement:
  * free list is a FIFO VecDeque seeded 0..num_blocks (:71), `allocate` pops
    the FRONT (:141), `free` pushes to the BACK (:199)  => ascending first
    use, FIFO reuse (the ":43 LIFO" comment in the reference is wrong);
  * `Block::decrement_ref` saturates at 0 (block.rs:152-155);
  * `free` on ref_count 0 removes the prefix-cache entry for the block's
    hash (:193-195);
  * `clear_prefix_cache` leaves Some(0) in every formerly cached block
    (:362-372), so a later `free` of such a block removes prefix_cache[0]
    if some other block has since been cached under hash 0 (quirk kept).
Pinned by the reference's own tests/block_manager_test.rs (ported in
tests/test_block_manager.py).
"""
from collections import deque


class OutOfBlocks(Exception):
    """Error::OutOfBlocks (src/error.rs:12-13)."""


class _Block:
    __slots__ = ("block_id", "block_size", "ref_count", "prefix_hash", "is_full")

    def __init__(self, block_id, block_size):
        self.block_id = block_id
        self.block_size = block_size
        self.ref_count = 1            # block.rs:96
        self.prefix_hash = None       # block.rs:97
        self.is_full = False


class BlockManagerOracle:
    def __init__(self, num_blocks, block_size):
        self.blocks = {}                              # :74
        self.free_list = deque(range(num_blocks))     # :71
        self.prefix_cache = {}                        # :76
        self.block_size = block_size
        self.num_blocks = num_blocks

    # :98-115
    def num_free_blocks(self):
        return len(self.free_list)

    def num_used_blocks(self):
        return len(self.blocks)

    def has_free_blocks(self):
        return len(self.free_list) > 0

    def can_allocate(self, n):
        return len(self.free_list) >= n

    # :140-147
    def allocate(self):
        if not self.free_list:
            raise OutOfBlocks()
        block_id = self.free_list.popleft()
        self.blocks[block_id] = _Block(block_id, self.block_size)
        return block_id

    # :162-173
    def allocate_many(self, n):
        if not self.can_allocate(n):
            raise OutOfBlocks()
        return [self.allocate() for _ in range(n)]

    # :187-204
    def free(self, block_id):
        blk = self.blocks.get(block_id)
        if blk is not None:
            blk.ref_count = max(blk.ref_count - 1, 0)
            if blk.ref_count == 0:
                if blk.prefix_hash is not None:
                    self.prefix_cache.pop(blk.prefix_hash, None)
                del self.blocks[block_id]
                self.free_list.append(block_id)
                return True
        return False

    # :215-217
    def free_many(self, ids):
        return sum(1 for i in ids if self.free(i))

    def ref_count(self, block_id):
        blk = self.blocks.get(block_id)
        return None if blk is None else blk.ref_count

    def prefix_hash_of(self, block_id):
        blk = self.blocks.get(block_id)
        return None if blk is None else blk.prefix_hash

    # :236-241
    def increment_ref(self, block_id):
        blk = self.blocks.get(block_id)
        if blk is None:
            return None
        blk.ref_count += 1
        return blk.ref_count

    # :251-255
    def decrement_ref(self, block_id):
        blk = self.blocks.get(block_id)
        if blk is None:
            return None
        blk.ref_count = max(blk.ref_count - 1, 0)
        return blk.ref_count

    # :292-304
    def get_cached_block(self, prefix_hash):
        block_id = self.prefix_cache.get(prefix_hash)
        if block_id is not None:
            if block_id in self.blocks:
                self.increment_ref(block_id)
                return block_id
            del self.prefix_cache[prefix_hash]
        return None

    # :312-317
    def cache_block(self, block_id, prefix_hash):
        blk = self.blocks.get(block_id)
        if blk is not None:
            blk.prefix_hash = prefix_hash
            self.prefix_cache[prefix_hash] = block_id

    # :337-347
    def allocate_with_prefix(self, prefix_hash):
        block_id = self.get_cached_block(prefix_hash)
        if block_id is not None:
            return block_id, True
        block_id = self.allocate()
        self.cache_block(block_id, prefix_hash)
        return block_id, False

    def is_prefix_cached(self, prefix_hash):
        return prefix_hash in self.prefix_cache

    def num_cached_prefixes(self):
        return len(self.prefix_cache)

    # :362-372
    def clear_prefix_cache(self):
        for _hash, block_id in self.prefix_cache.items():
            blk = self.blocks.get(block_id)
            if blk is not None:
                blk.prefix_hash = 0
        self.prefix_cache.clear()

    # :377-382
    def reset(self):
        self.blocks.clear()
        self.prefix_cache.clear()
        self.free_list.clear()
        self.free_list.extend(range(self.num_blocks))

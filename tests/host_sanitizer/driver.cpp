// Stress driver of the host half of the C ABI under AddressSanitizer + UBSan (no GPU, no device
// code): block tables, the allocator with its prefix cache, hashing, the RoPE table, the slot
// resolver, batch metadata with its roll-back, the scheduler over long random traces, and every
// argument-validation path of the device entries (which must fail before touching a device).
// Exit code 0 = no sanitizer report and every self-check held.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "nvllm_attn.h"

#define REQUIRE(x)                                                          \
    do {                                                                    \
        if (!(x)) {                                                         \
            std::fprintf(stderr, "REQUIRE failed: %s (%s:%d)\n", #x, __FILE__, __LINE__); \
            std::exit(1);                                                   \
        }                                                                   \
    } while (0)

static void block_tables() {
    nvllm_block_table *t = nvllm_block_table_new(16);
    REQUIRE(t && nvllm_block_table_is_empty(t));
    nvllm_block_table_append_block(t, 5);
    nvllm_block_table_append_block(t, 12);
    int64_t slots[32];
    REQUIRE(nvllm_block_table_get_slot_mapping(t, 20, slots, 32) == 20);   // block_test.rs:75-90
    REQUIRE(slots[0] == 80 && slots[15] == 95 && slots[16] == 192 && slots[19] == 195);
    REQUIRE(nvllm_block_table_get_slot_mapping(t, 40, slots, 32) == 32);   // past the table: dropped
    REQUIRE(nvllm_block_table_get_slot_mapping(t, 40, slots, 7) == 32);    // cap < length: no overrun
    int64_t id = -1;
    REQUIRE(nvllm_block_table_get_block_id(t, 1, &id) == NVLLM_OK && id == 12);
    REQUIRE(nvllm_block_table_get_block_id(t, 2, &id) == NVLLM_ERR_BLOCK_INDEX_OOB);
    REQUIRE(nvllm_block_table_get_block_id(t, -1, &id) == NVLLM_ERR_BLOCK_INDEX_OOB);
    int64_t ids[1];
    REQUIRE(nvllm_block_table_get_physical_block_ids(t, ids, 1) == 2 && ids[0] == 5);
    nvllm_block_table_clear(t);
    REQUIRE(nvllm_block_table_num_blocks(t) == 0);
    nvllm_block_table_free(t);
    REQUIRE(nvllm_block_table_new(0) == nullptr);
    REQUIRE(nvllm_compute_num_blocks(17, 16) == 2 && nvllm_compute_num_blocks(0, 16) == 0);
}

static void allocator(std::mt19937_64 &rng) {
    nvllm_block_manager *m = nvllm_block_manager_new(64, 16);
    std::vector<int64_t> live;
    for (int step = 0; step < 20000; ++step) {
        const int op = (int)(rng() % 7);
        int64_t id = -1;
        if (op == 0) {
            if (nvllm_block_manager_allocate(m, &id) == NVLLM_OK) live.push_back(id);
            else REQUIRE(nvllm_block_manager_num_free_blocks(m) == 0);
        } else if (op == 1 && !live.empty()) {
            const size_t i = rng() % live.size();
            nvllm_block_manager_free(m, live[i]);
            live.erase(live.begin() + (long)i);
        } else if (op == 2) {
            int cached = 0;
            if (nvllm_block_manager_allocate_with_prefix(m, rng() % 9, &id, &cached) == NVLLM_OK) live.push_back(id);
        } else if (op == 3 && !live.empty()) {
            const int64_t x = live[rng() % live.size()];
            if (nvllm_block_manager_increment_ref(m, x) > 0) live.push_back(x);
        } else if (op == 4) {
            int64_t many[8];
            const int64_t n = (int64_t)(rng() % 8);
            if (nvllm_block_manager_allocate_many(m, n, many) == NVLLM_OK)
                for (int64_t i = 0; i < n; ++i) live.push_back(many[i]);
        } else if (op == 5) {
            uint64_t h = 0;
            nvllm_block_manager_prefix_hash(m, (int64_t)(rng() % 80) - 8, &h);   // out-of-range ids too
            nvllm_block_manager_ref_count(m, (int64_t)(rng() % 80) - 8);
            nvllm_block_manager_free(m, 64 + (int64_t)(rng() % 4));
            nvllm_block_manager_decrement_ref(m, -3);
        } else if (op == 6 && step % 1000 == 999) {
            nvllm_block_manager_clear_prefix_cache(m);
        }
        REQUIRE(nvllm_block_manager_num_free_blocks(m) + nvllm_block_manager_num_used_blocks(m) == 64);
    }
    nvllm_block_manager_reset(m);
    REQUIRE(nvllm_block_manager_num_free_blocks(m) == 64 && nvllm_block_manager_num_cached_prefixes(m) == 0);
    nvllm_block_manager_free_handle(m);
    REQUIRE(nvllm_block_manager_new(-1, 16) == nullptr);
}

static void hashing_rope_slots(std::mt19937_64 &rng) {
    uint32_t toks[16];
    for (int i = 0; i < 16; ++i) toks[i] = (uint32_t)i;
    const uint64_t a = nvllm_hash_token_block(toks, 16, 0, 0), b = nvllm_hash_token_block(toks, 16, 1, 999);
    REQUIRE(a != b && a == nvllm_hash_token_block(toks, 16, 0, 7));
    nvllm_hash_token_block(toks, 0, 0, 0);
    nvllm_hash_token_block(nullptr, 0, 1, 5);
    for (int n = 0; n < 16; ++n) nvllm_hash_token_block(toks, n, n & 1, rng());   // every tail length
    std::vector<float> cosv(300 * 128), sinv(300 * 128);
    REQUIRE(nvllm_rope_cache_f32(128, 300, 1e6, cosv.data(), sinv.data()) == NVLLM_OK);
    REQUIRE(cosv[0] == 1.0f && sinv[0] == 0.0f && cosv[128 + 3] == cosv[128 + 64 + 3]);
    REQUIRE(nvllm_rope_cache_f32(5, 8, 1e4, cosv.data(), sinv.data()) == NVLLM_ERR_CONFIG);
    REQUIRE(nvllm_rope_cache_f32(4, 0, 1e4, cosv.data(), sinv.data()) == NVLLM_OK);
    std::vector<int64_t> s(1000), out(1000);
    for (auto &x : s) x = (int64_t)(rng() % 200) - 5;
    const int64_t sup = nvllm_slot_mapping_last_writer(s.data(), 1000, out.data());
    REQUIRE(sup > 0 && nvllm_slot_mapping_last_writer(s.data(), 0, out.data()) == 0);
    REQUIRE(nvllm_kv_cache_layer_bytes(16, 4, 8, 64, 1) == 131072);               // kv_cache_test.rs:27
    REQUIRE(nvllm_kv_cache_total_bytes(16, 4, 8, 64, 2, 1) == 524288);
}

static void batches() {
    nvllm_batch *b = nvllm_batch_new(4);
    nvllm_block_table *t = nvllm_block_table_new(16);
    for (int i = 0; i < 3; ++i) nvllm_block_table_append_block(t, 10 + i);
    REQUIRE(nvllm_batch_add_sequence(b, t, 40, 8) == NVLLM_OK);
    REQUIRE(nvllm_batch_add_sequence(b, t, 49, 1) == NVLLM_ERR_CONFIG);           // beyond the blocks
    REQUIRE(nvllm_batch_add_sequence(b, t, 4, 8) == NVLLM_ERR_CONFIG);            // more new than context
    nvllm_block_table_append_block(t, (int64_t)1 << 40);                          // id beyond int32
    REQUIRE(nvllm_batch_add_sequence(b, t, 40, 1) == NVLLM_ERR_CONFIG);
    REQUIRE(nvllm_batch_num_seqs(b) == 1 && nvllm_batch_num_tokens(b) == 8);      // rejected: untouched
    REQUIRE(nvllm_batch_query_start(b)[1] == 8 && nvllm_batch_slot_mapping(b)[0] == 12 * 16);
    nvllm_block_table_append_block(t, 3);                                         // 5 blocks > max 4
    REQUIRE(nvllm_batch_add_sequence(b, t, 40, 1) == NVLLM_ERR_CONFIG);
    REQUIRE(nvllm_batch_upload(b, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) != NVLLM_ERR_NULL);
    nvllm_batch_clear(b);
    REQUIRE(nvllm_batch_num_seqs(b) == 0);
    nvllm_batch_free(b);
    nvllm_block_table_free(t);
    REQUIRE(nvllm_batch_new(0) == nullptr);
}

static void scheduler_traces(std::mt19937_64 &rng) {
    for (int variant = 0; variant < 8; ++variant) {
        nvllm_scheduler_config c;
        nvllm_scheduler_config_default(&c);
        c.max_num_seqs = 2 + (int64_t)(rng() % 7);
        c.max_prefill_tokens = 20 + (int64_t)(rng() % 200);
        c.enable_chunked_prefill = variant & 1;
        c.chunk_size = 8 + (int64_t)(rng() % 56);
        c.enable_preemption = (variant >> 1) & 1;
        c.reschedule_partial_prefill = (variant >> 2) & 1;
        c.enable_prefix_caching = (variant >> 2) & 1;
        nvllm_scheduler *s = nvllm_scheduler_new(&c, 16, 6 + (int64_t)(rng() % 40));
        REQUIRE(s);
        nvllm_batch *batch = nvllm_batch_new(64);
        std::vector<uint64_t> live;
        uint64_t next_id = 1;
        for (int step = 0; step < 1500; ++step) {
            const unsigned r = (unsigned)(rng() % 100);
            if (r < 30) {
                std::vector<uint32_t> prompt(1 + rng() % 90);
                for (auto &x : prompt) x = (uint32_t)(rng() % 50);   // small vocabulary: shared prefixes happen
                REQUIRE(nvllm_scheduler_add_sequence(s, next_id, prompt.data(), (int64_t)prompt.size(),
                                                     (int32_t)(rng() % 7) - 3) == NVLLM_OK);
                live.push_back(next_id++);
            } else if (r < 75) {
                REQUIRE(nvllm_scheduler_schedule(s) == NVLLM_OK);
                const int64_t np = nvllm_scheduler_out_num_prefill(s), nd = nvllm_scheduler_out_num_decode(s);
                const uint64_t *pid = nvllm_scheduler_out_prefill_ids(s);
                const int64_t *chunk = nvllm_scheduler_out_prefill_chunk_sizes(s);
                nvllm_batch_clear(batch);
                if (nd) nvllm_scheduler_fill_decode_batch(s, batch);
                uint64_t added[64];
                if (np) nvllm_scheduler_fill_prefill_batch(s, chunk[0], batch, added, 64);
                std::vector<uint64_t> pre(pid, pid + np);
                std::vector<int64_t> ch(chunk, chunk + np);
                for (int64_t i = 0; i < np; ++i) nvllm_scheduler_mark_prefilled(s, pre[(size_t)i], ch[(size_t)i]);
                const uint64_t *did = nvllm_scheduler_out_decode_ids(s);
                std::vector<uint64_t> dec(did, did + nd);
                for (uint64_t id : dec) nvllm_scheduler_append_token(s, id, 7);
            } else if (r < 87 && !live.empty()) {
                nvllm_scheduler_finish_sequence(s, live[rng() % live.size()], (int)(rng() % 4));
            } else if (!live.empty()) {
                const size_t i = rng() % live.size();
                nvllm_scheduler_remove_sequence(s, live[i]);
                live.erase(live.begin() + (long)i);
            }
            if (!live.empty()) {
                const uint64_t id = live[rng() % live.size()];
                int64_t blocks[64];
                uint32_t toks[8];
                uint64_t run[4];
                nvllm_scheduler_seq_block_ids(s, id, blocks, 64);
                nvllm_scheduler_seq_output_tokens(s, id, toks, 8);
                nvllm_scheduler_running_sequence_ids(s, run, 4);
                for (int w = 0; w < 10; ++w) nvllm_scheduler_seq_info(s, id, w);
                nvllm_scheduler_seq_status(s, id + 1000);
                nvllm_scheduler_seq_set_status(s, id, (int)(rng() % 4));
            }
            REQUIRE(nvllm_scheduler_num_free_blocks(s) >= 0);
        }
        nvllm_scheduler_reset(s);
        REQUIRE(nvllm_scheduler_num_running(s) == 0 && nvllm_scheduler_num_used_blocks(s) == 0);
        nvllm_batch_free(batch);
        nvllm_scheduler_free(s);
    }
}

static void device_entries_validate_first() {
    char buf[64];
    void *fake = buf;   // never dereferenced: every call below must fail in validation or at "no device"
    REQUIRE(nvllm_flash_attn_fwd_bf16(nullptr, fake, fake, fake, nullptr, 1, 1, 1, 1, 1, 128, 1.f, 1, nullptr) == NVLLM_ERR_NULL);
    REQUIRE(nvllm_flash_attn_fwd_bf16(fake, fake, fake, fake, nullptr, 1, 1, 1, 3, 2, 128, 1.f, 1, nullptr) == NVLLM_ERR_CONFIG);
    REQUIRE(nvllm_flash_attn_fwd_f32(fake, fake, fake, fake, nullptr, 1, 1, 1, 2, 2, 512, 1.f, 1, nullptr) == NVLLM_ERR_UNSUPPORTED);
    REQUIRE(nvllm_flash_attn_fwd_f32(fake, fake, fake, fake, nullptr, 0, 1, 1, 2, 2, 64, 1.f, 1, nullptr) == NVLLM_OK);
    REQUIRE(nvllm_paged_attn_bf16(fake, fake, fake, (const int32_t *)fake, (const int32_t *)fake, fake, nullptr, 0,
                                  7, 1, 32, 8, 128, 16, 128, 64, 0.1f, nullptr) == NVLLM_ERR_CONFIG);   // workspace
    REQUIRE(std::strstr(nvllm_get_last_error(), "needs 4096") != nullptr);
    REQUIRE(nvllm_paged_attn_f32(fake, fake, fake, (const int32_t *)fake, (const int32_t *)fake, fake, nullptr, 0,
                                 1, 1, 4, 2, 8, 4, 0, 4, 0.1f, nullptr) == NVLLM_ERR_CONFIG);           // empty table
    REQUIRE(nvllm_paged_attn_bf16_noalloc(fake, fake, fake, (const int32_t *)fake, (const int32_t *)fake, fake, 2, 1,
                                          32, 8, 128, 16, 4, 64, 0.1f, nullptr) == NVLLM_ERR_HIP);      // no device
    REQUIRE(nvllm_reshape_and_cache_bf16(fake, fake, fake, fake, nullptr, 4, 8, 128, 16, 4, nullptr) == NVLLM_ERR_NULL);
    REQUIRE(nvllm_gather_cache_blocks(fake, (const int32_t *)fake, 0, fake, 4, 16, 8, 128, 0, nullptr) == NVLLM_ERR_CONFIG);
    REQUIRE(nvllm_qk_norm_rope_cache_bf16(fake, fake, fake, fake, nullptr, fake, fake, 1e-6f, fake, fake, 64,
                                          (const int32_t *)fake, fake, fake, (const int64_t *)fake, 4, 4, 2, 127,
                                          16, 4, nullptr) == NVLLM_ERR_CONFIG);                          // odd dim
    nvllm_block_table *t = nvllm_block_table_new(4);
    REQUIRE(nvllm_paged_attention_seq(fake, 1, 1, fake, fake, 4, t, 4, 4, 2, 8, 0.3, 1, fake, fake, 64, nullptr) ==
            NVLLM_ERR_CONFIG);                                                                          // empty table
    nvllm_block_table_append_block(t, 9);
    REQUIRE(nvllm_paged_attention_seq(fake, 1, 1, fake, fake, 4, t, 4, 4, 2, 8, 0.3, 1, fake, fake, 64, nullptr) ==
            NVLLM_ERR_TENSOR);                                                                          // id >= pool
    nvllm_block_table_free(t);
    int64_t slots[3] = {0, 1, 99};
    REQUIRE(nvllm_write_kv_to_cache(fake, fake, 3, fake, fake, 2, slots, 2, 4, 2, 4, 1, fake, 64, nullptr) == NVLLM_ERR_CONFIG);
    REQUIRE(nvllm_write_kv_to_cache(fake, fake, 3, fake, fake, 2, slots, 3, 4, 2, 4, 1, fake, 64, nullptr) == NVLLM_ERR_TENSOR);
    REQUIRE(nvllm_debug_set_policy("NVLLM_PAGED_SPLITS", 3) == NVLLM_OK);
    REQUIRE(nvllm_debug_set_policy("nope", 3) == NVLLM_ERR_CONFIG && nvllm_debug_set_policy(nullptr, 0) == NVLLM_ERR_CONFIG);
    nvllm_debug_reset_policy();
}

int main() {
    std::mt19937_64 rng(20261111);
    block_tables();
    allocator(rng);
    hashing_rope_slots(rng);
    batches();
    scheduler_traces(rng);
    device_entries_validate_first();
    std::puts("host sanitizer driver: ok");
    return 0;
}

/*
 * brisk_hip.h -- C-ABI of the MI355X-native Brisk hot path (libbrisk_hip.so).
 *
 * The reference has no FFI layer: its boundary is the header-only C++ API
 * consumed by apps/counter.cpp.  This C-ABI is the boundary
 * between that host-side C++ (brisk_amd/include/Brisk.hpp, same names as the
 * reference's) and the HIP kernels.  Plain pointers and sizes only; no C++,
 * torch or HIP types appear in any signature.  Every entry point returns an
 * int status (BRISK_HIP_OK == 0) and never throws.
 *
 * Threads.  The reference lets any number of OpenMP threads call insert_superkmer /
 * get_superkmer at once under its lock stripes (brisk/Brisk.hpp:112-114,140-143,
 * brisk/DenseMenuYo.hpp:110-118; apps/counter.cpp:197-227,314-346).  Here the unit of
 * work is a batch, and a handle takes ONE call at a time (a per-handle lock inside the
 * library; one HIP stream per handle): any number of host threads may call any entry
 * points on one handle concurrently -- e.g. one thread streaming brisk_hip_insert_reads
 * batches while another issues brisk_hip_get_reads / brisk_hip_lookup -- and every call
 * runs against the index as left by a whole number of the other threads' completed
 * calls.  A concurrent get therefore observes whole batches only: never a partly
 * inserted batch, never a torn entry (a key without its count, a count updated for some
 * of a read's k-mers only).  Which batches it observes depends on timing, as in the
 * reference.  Calls on different handles are independent.  brisk_hip_last_error is the
 * one exception: it returns the handle's last message without taking the lock.
 *
 * Reference interface each entry point stands in for (file:line under the
 * reference tree):
 *   brisk_hip_create            Parameters ctor + Brisk ctor + DenseMenuYo ctor
 *                               (brisk/parameters.hpp:24-34, brisk/Brisk.hpp:47-52,
 *                                brisk/DenseMenuYo.hpp:104-138)
 *   brisk_hip_destroy           Brisk dtor (brisk/Brisk.hpp:57-59)
 *   brisk_hip_clear             `delete menu; menu = new DenseMenuYo` (what Brisk::reallocate does
 *                               to start over, brisk/Brisk.hpp:220-222): an empty index that keeps
 *                               its device memory reserved
 *   brisk_hip_reallocate        Brisk::reallocate (brisk/Brisk.hpp:202-224)
 *   brisk_hip_insert_reads      count_sequence loop: SuperKmerEnumerator::next +
 *                               Brisk::protect_data/insert_superkmer/unprotect_data +
 *                               the counter update (apps/counter.cpp:231-276,
 *                               brisk/Kmers.cpp:522-603, brisk/Brisk.hpp:123-161);
 *                               also the declared-but-undefined
 *                               Brisk::insert_sequence (brisk/Brisk.hpp:27)
 *   brisk_hip_get_reads / brisk_hip_get_packed
 *                               query_sequence (apps/counter.cpp:281-310) over
 *                               Brisk::get_superkmer (brisk/Brisk.hpp:102-118) /
 *                               Brisk::get_sequence (brisk/Brisk.hpp:28); _packed: reads and sums on the device
 *   brisk_hip_lookup            Brisk::get (brisk/Brisk.hpp:64-69)
 *   brisk_hip_enumerate         Brisk::next / restart_kmer_enumeration
 *                               (brisk/Brisk.hpp:166-179, brisk/DenseMenuYo.hpp:476-521)
 *   brisk_hip_stats             Brisk::stats (brisk/Brisk.hpp:194-197,
 *                               brisk/DenseMenuYo.hpp:545-568)
 *   brisk_hip_memory_info / brisk_hip_insert_slack   no reference counterpart (Brisk::stats reports the process' peak RSS): the arena's bookkeeping
 *   brisk_hip_checksum          the next()+get() walk of verif_counts (apps/counter.cpp:90-126), reduced
 *                               to a digest on the device
 *   brisk_hip_scan_packed / brisk_hip_route_records / brisk_hip_insert_records
 *                               the same insert path cut at the super-k-mer
 *                               boundary (the vector<kmer_full> handed from
 *                               SuperKmerEnumerator::next to Brisk::insert_superkmer,
 *                               apps/counter.cpp:242-261) so that records can be
 *                               exchanged between GPUs (no reference counterpart:
 *                               the reference is single-process)
 *   brisk_hip_set_owner_cuts    no reference counterpart: which partition range each GPU of a sharded job owns
 *   brisk_hip_export_hist / brisk_hip_export_hist_add / brisk_hip_insert_records_hist
 *                               no reference counterpart: the per-partition record counts of a scan travel with
 *                               the records so that the owner need not count them again
 *   brisk_hip_scan_query / brisk_hip_route_tagged / brisk_hip_query_records
 *                               the query path (apps/counter.cpp:281-310, brisk/Brisk.hpp:102-118) cut at the
 *                               same boundary, for a get across bucket-range shards
 *   brisk_hip_scan_sequence     SuperKmerEnumerator ctor + next() until empty (brisk/Kmers.cpp:509-603)
 *   brisk_hip_upsert_kmers      Brisk::insert_superkmer (brisk/Brisk.hpp:123-147) minus the DATA pointers,
 *                               which the facade forms from the returned ids
 *   brisk_hip_find_kmers        Brisk::get_superkmer / Brisk::get (brisk/Brisk.hpp:64-69,102-118)
 *   brisk_hip_enumerate_ids     Brisk::next (brisk/Brisk.hpp:166-172)
 *   brisk_hip_pack_ascii        nuc2int (brisk/Kmers.cpp:442-444) applied in bulk
 *   brisk_hip_synth_reads       no reference counterpart (benchmark input,
 *                               SURVEY.md 8(d))
 *   brisk_hip_debug_order_keys  bfc_hash_64 + DecyclingSet::memDouble in bulk
 *                               (brisk/hashing.cpp:8-19, brisk/Decycling.cpp:38-52); test hook
 */
#ifndef BRISK_HIP_H
#define BRISK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BRISK_HIP_ABI_VERSION 4

enum {
    BRISK_HIP_OK = 0,
    BRISK_HIP_EINVAL = 1,       /* parameter contract violated (e.g. b > m, m even) */
    BRISK_HIP_EUNSUPPORTED = 2, /* valid for the reference but outside this library's envelope */
    BRISK_HIP_EHIP = 3,         /* a HIP runtime call failed; see brisk_hip_last_error */
    BRISK_HIP_ENOMEM = 4,       /* device or host allocation failed */
    BRISK_HIP_ECAPACITY = 5,    /* caller-provided output buffer too small */
    BRISK_HIP_ENODEVICE = 6     /* no usable gfx950 device */
};

typedef struct brisk_hip_index brisk_hip_index;

typedef struct brisk_hip_options {
    uint32_t struct_size;       /* sizeof(brisk_hip_options), for ABI growth */
    int32_t device;             /* HIP device ordinal */
    void *stream;               /* hipStream_t to run on, NULL: the library creates one */
    uint32_t part_bits;         /* log2(#partitions), at most 2b; 0: default (2^24, also when 2b < 24: see brisk_hip_layout.ext_bits).
                                 * The insert is at its best with 10-20 records (a few hundred k-mer instances) per partition
                                 * and call: 2^24 suits batches of 50 M reads on one index; a sharded job whose owners each
                                 * receive that much (N x 50 M reads per batch over N owners) wants 24 + log2(N); a job of ONE
                                 * smaller batch of short k-mers wants fewer (k31 m15 b14, 20 M reads: 2^22 -- two-word records of
                                 * nine k-mers leave 2^24 partitions with a dozen entries each; the k63 insert loses with fewer
                                 * than 2^24).  brisk_amd.exchange.suggest_part_bits states the rule bench.py uses */
    uint32_t owner_rank;        /* this process' rank among n_owners bucket-range owners */
    uint32_t n_owners;          /* 0 or 1: this index owns every bucket; at most 256 (else EUNSUPPORTED) */
    uint64_t arena_entries;     /* initial entry capacity of the k-mer arena; 0: grow on demand */
    uint64_t max_batch_reads;   /* reads per internal scan batch; 0: default */
    uint32_t entry_ids;         /* 1: entry-id mode (per-call facade API): every entry gets a stable dense id in
                                 * insertion order and DATA lives with the caller, indexed by id; bulk count
                                 * entry points are refused on such an index */
    uint32_t immediate_inserts; /* 0: an insert call whose batch would leave the partitions nearly empty (fewer than ~24 M reads
                                 * at k=63 on 2^24 partitions) is scanned at once and its records inserted together with those of
                                 * the next such calls -- at the latest when any call other than an insert arrives (they all
                                 * complete pending inserts first), so no call ever sees an index without them; an error of
                                 * the deferred part (BRISK_HIP_ENOMEM) is returned by the call that completes it.  Streams
                                 * of 2 M-read batches go in 3x faster that way.  1: every insert call completes before it
                                 * returns. */
} brisk_hip_options;

/* ---- lifetime ----------------------------------------------------------- */
/* coef_table: the 4*m doubles of DecyclingSet(m) computed on the HOST with libm
 * (brisk/Decycling.cpp:7-13); the device never evaluates sin().  data_bytes must
 * be 1 (the counter app's uint8_t DATA). */
int brisk_hip_create(brisk_hip_index **out, uint8_t k, uint8_t m, uint8_t b, uint32_t data_bytes,
                     const double *coef_table, const brisk_hip_options *opt);
int brisk_hip_destroy(brisk_hip_index *h);
/* back to the empty index; device memory stays reserved for the next job */
int brisk_hip_clear(brisk_hip_index *h);
const char *brisk_hip_last_error(const brisk_hip_index *h);
int brisk_hip_sync(brisk_hip_index *h);
uint32_t brisk_hip_abi_version(void);

/* derived constants (mirrors brisk/parameters.hpp:24-34) */
typedef struct brisk_hip_layout {
    uint32_t k, m, b, m_reduc, compacted_size, allocated_bytes;
    uint32_t record_words;      /* u64 words per super-k-mer record (incl. header word) */
    uint32_t part_bits;         /* log2(#partitions) */
    uint32_t n_owners, owner_rank;
    uint32_t ext_bits;          /* a record's routing id (header bits 0..31) = bucket id << ext_bits | ext_bits more bits: first
                                 * more of the same hashed minimizer, then (lowest) cls_bits of class; 0 when 2b >= 24 or
                                 * part_bits was given: the bucket id itself */
    uint32_t cls_bits;          /* 2m < 24 only: the lowest cls_bits of a routing id are min(minimizer_idx / cls_width,
                                 * 2^cls_bits - 1) of the record's k-mers, and a super-k-mer spanning several classes is
                                 * scanned into one record per class (each a valid super-k-mer of the same bucket) */
    uint32_t cls_width;
} brisk_hip_layout;
int brisk_hip_get_layout(const brisk_hip_index *h, brisk_hip_layout *out);

/* ---- bulk count (DATA = uint8_t counter: first touch = 1, then ++ mod 256) -- */
/* HOST buffers: `bases` = concatenated sequences, `offsets[n_reads+1]`.  Sequences
 * must be clean ([ACGTacgt]; the N-splitting of counter.cpp:130-169 is the caller's);
 * sequences shorter than k are skipped (counter.cpp:233-235).  A large batch is taken in
 * pieces: a dozen host threads turn the bytes into the 2-bit stream (nuc2int, Kmers.cpp:442-444)
 * while they stage them into pinned memory, and the scan of one piece runs under the upload of
 * the next (environment: BRISK_UPLOAD_LANES, BRISK_PIPE_PIECES, BRISK_UPLOAD_PIPELINE=0,
 * BRISK_HOST_PACK=0 for ASCII over PCIe and k_pack_ascii on arrival). */
int brisk_hip_insert_reads(brisk_hip_index *h, const char *bases, const uint64_t *offsets, uint64_t n_reads);

/* DEVICE buffers: 2-bit packed stream (16 nts per u32, first nt in the top bits;
 * A0 C1 T2 G3) and per-read start offsets in nucleotides, starts[n_reads+1].
 * d_packed must be readable 8 bytes past the last used word. */
int brisk_hip_insert_packed(brisk_hip_index *h, const uint32_t *d_packed, const uint64_t *d_starts, uint64_t n_reads);

/* ---- bulk query ---------------------------------------------------------- */
/* per_read_sum[r] = sum of the counts of the read's k-mers that are present,
 * with query_sequence's quirk: enumeration of a read stops at the first
 * super-k-mer after the first whose minimizer value is 0 (counter.cpp:304-306). */
int brisk_hip_get_reads(brisk_hip_index *h, const char *bases, const uint64_t *offsets, uint64_t n_reads,
                        uint64_t *per_read_sum);
/* the same with reads and sums resident on the device (layout of brisk_hip_insert_packed; d_per_read_sum[n_reads]) */
int brisk_hip_get_packed(brisk_hip_index *h, const uint32_t *d_packed, const uint64_t *d_starts, uint64_t n_reads,
                         uint64_t *d_per_read_sum);

/* point lookups of UNHASHED (kmer_s, minimizer_idx) pairs, as Brisk::get takes them.
 * HOST arrays; out_found[i] in {0,1}; out_data[i] valid when found. */
int brisk_hip_lookup(brisk_hip_index *h, const uint64_t *kmer_lo, const uint64_t *kmer_hi, const uint8_t *minimizer_idx,
                     uint64_t n, uint8_t *out_data, uint8_t *out_found);

/* ---- enumeration --------------------------------------------------------- */
/* Walks the index in ascending partition (bucket-range) order, storage order
 * inside one.  *cursor = 0 restarts (restart_kmer_enumeration); the call
 * returns up to cap entries into HOST arrays (k-mers unhashed, as Brisk::next
 * yields them), advances *cursor, and sets *n_out; *n_out == 0 means done. */
int brisk_hip_enumerate(brisk_hip_index *h, uint64_t *cursor, uint64_t *out_lo, uint64_t *out_hi,
                        uint8_t *out_minimizer_idx, uint8_t *out_data, uint64_t cap, uint64_t *n_out);

/* nb_buckets and nb_kmers are exact and order independent; nb_skmers and
 * largest_bucket depend on insertion order in the reference and are reported
 * here as: super-k-mer records received, largest partition (entries). */
int brisk_hip_stats(brisk_hip_index *h, uint64_t *nb_buckets, uint64_t *nb_skmers, uint64_t *nb_kmers,
                    uint64_t *memory_bytes, uint64_t *largest_bucket);

/* Brisk::reallocate (brisk/Brisk.hpp:202-224: the index re-bucketed to (m + 2, b + 2); dormant in the reference): every
 * entry of `from` goes into `to` -- an EMPTY bulk-count index over the same k and device, created by the caller with the
 * new (m, b) -- under the identity (kmer_s, minimizer_idx) that SuperKmerEnumerator gives the k-mer at the new m, with
 * its count; entries that the new minimizer maps to one identity merge, counts added mod 256.  `from` stays as it is. */
int brisk_hip_reallocate(brisk_hip_index *from, brisk_hip_index *to);

/* Where the arena's memory is (no reference counterpart; Brisk::stats reports the process' peak RSS, brisk/Brisk.hpp:184-189):
 * out[0] = device memory mapped behind this index's arena, out[1] = virtual address range this index has reserved for it
 * (0 without virtual memory management), out[2] = device memory held, process-wide, by the pooled arenas of destroyed
 * indexes (handed to the next index, given back when an allocation fails for lack of memory), out[3] = address space,
 * process-wide, that retired arenas keep reserved for the life of the process (no memory behind it). */
int brisk_hip_memory_info(brisk_hip_index *h, uint64_t out[4]);
/* Entries of arena the single-pass insert sets aside beyond a batch's own need ("every k-mer instance is new"): one partly used
 * private chunk per persistent insert wave (resident waves x chunk entries).  A batch's insert needs room for
 * instances + instances / 7 + this many entries, or it is split in halves (BRISK_HIP_ENOMEM when even one read does not fit). */
int brisk_hip_insert_slack(brisk_hip_index *h, uint64_t *entries);

/* Order-independent digest of the whole index, for parity checks at sizes where the multiset
 * cannot be compared line by line: out[0] = number of entries, out[1] = sum of counts,
 * out[2] = sum over entries of mix(kmer_lo, kmer_hi, minimizer_idx, count) mod 2^64 with
 * mix(a,b,c,d) = f(a ^ f(b ^ f(c << 8 | d))), f = the splitmix64 finaliser (k-mers unhashed,
 * as Brisk::next yields them).  Digests of bucket-range shards add up to the whole index's. */
int brisk_hip_checksum(brisk_hip_index *h, uint64_t out[3]);

/* ---- the path cut at the super-k-mer boundary (multi-GPU) ----------------- */
/* scan: d_records receives up to cap_records records of record_words u64 each;
 * *n_records (HOST) receives the count.  BRISK_HIP_ECAPACITY if cap is too small
 * (nothing is inserted by a scan, so the call can simply be repeated). */
int brisk_hip_scan_packed(brisk_hip_index *h, const uint32_t *d_packed, const uint64_t *d_starts, uint64_t n_reads,
                          uint64_t *d_records, uint64_t cap_records, uint64_t *n_records);
/* upper bound on the records a scan of these reads can emit (HOST result) */
int brisk_hip_scan_bound(brisk_hip_index *h, const uint64_t *d_starts, uint64_t n_reads, uint64_t *bound);
/* route: reorder records so that each owner's records are contiguous, owner 0
 * first; counts[n_owners] (HOST) receives records per owner.  d_out may not alias d_in. */
int brisk_hip_route_records(brisk_hip_index *h, const uint64_t *d_records, uint64_t n_records,
                            uint64_t *d_out, uint64_t *counts);
/* insert records whose buckets this index owns */
int brisk_hip_insert_records(brisk_hip_index *h, const uint64_t *d_records, uint64_t n_records);
/* The scan of a sharded index also counts its records per bucket-range partition (records in the low,
 * k-mer instances in the high 32 bits).  export_hist copies that histogram (2^part_bits u64, partition
 * order = owner order) to d_hist_out right after brisk_hip_scan_packed; partitions_per_owner[n_owners]
 * (HOST) receives the length of each owner's slice, so that the slices can travel with the records.
 * insert_records_hist is insert_records for an owner that received one such slice of ITS range from
 * each scanning rank (n_slices slices of equal length, back to back): it adds them up instead of
 * counting the received records again.  export_hist_add ADDS the histogram to d_hist_acc (2^part_bits u64 the
 * caller zeroed) instead: a rank that scans its reads in several pieces sends one summed slice per owner per
 * batch, not one per piece. */
int brisk_hip_export_hist(brisk_hip_index *h, uint64_t *d_hist_out, uint64_t *partitions_per_owner);
int brisk_hip_export_hist_add(brisk_hip_index *h, uint64_t *d_hist_acc, uint64_t *partitions_per_owner);
int brisk_hip_insert_records_hist(brisk_hip_index *h, const uint64_t *d_records, uint64_t n_records,
                                  const uint64_t *d_hist_slices, uint32_t n_slices);
/* Ownership of a sharded index (n_owners > 1): owner o holds the contiguous partition range [first_partition[o],
 * first_partition[o + 1]) (HOST array of n_owners + 1 entries, first_partition[0] = 0, first_partition[n_owners] = 2^part_bits,
 * ascending; an owner may hold nothing).  By default the ranges are equal (owner = partition * N >> part_bits).  Equal ranges do
 * not carry equal loads -- a partition is a range of minimizer-hash values and minimizers are the smallest hashes of their
 * windows ("histogram-balanced cut points if skew > 1.3x") -- so a job takes its cut points from the
 * partition histogram of a first scan (export_hist, all ranks' histograms added up) and hands the SAME array to every rank
 * before anything is routed or inserted: EINVAL on an index that holds entries.  No reference counterpart (single process). */
int brisk_hip_set_owner_cuts(brisk_hip_index *h, const uint64_t *first_partition);
/* The query path cut at the same boundary.  scan_query = the scan as query_sequence runs it (a read's
 * enumeration stops at the first super-k-mer after the first whose returned minimizer is 0,
 * apps/counter.cpp:304-306); d_tags[i] = index of the read record i came from.  route_tagged = route_records
 * carrying the tags along.  query_records: d_sums[i] = sum of the counts of record i's k-mers present in
 * THIS index (records whose buckets it owns), what Brisk::get_superkmer + the caller's loop add up
 * (brisk/Brisk.hpp:102-118, apps/counter.cpp:296-303).  A sharded get is scan_query -> route_tagged ->
 * all-to-all -> query_records on the owner -> all-to-all back -> per_read[tag] += sum. */
int brisk_hip_scan_query(brisk_hip_index *h, const uint32_t *d_packed, const uint64_t *d_starts, uint64_t n_reads,
                         uint64_t *d_records, uint32_t *d_tags, uint64_t cap_records, uint64_t *n_records);
int brisk_hip_route_tagged(brisk_hip_index *h, const uint64_t *d_records, const uint32_t *d_tags, uint64_t n_records,
                           uint64_t *d_out, uint32_t *d_tags_out, uint64_t *counts);
int brisk_hip_query_records(brisk_hip_index *h, const uint64_t *d_records, uint64_t n_records, uint64_t *d_sums);

/* ---- the per-call API under the C++ facade (entry-id mode) --------------------- */
/* SuperKmerEnumerator over one clean sequence (len >= k): every vector next() would
 * return, in order.  HOST arrays: skm_ret/skm_n sized >= len-k+1 vectors, km_* sized
 * >= cap_kmers >= len-k+1 k-mers (k-mers unhashed, vector after vector). */
int brisk_hip_scan_sequence(brisk_hip_index *h, const char *bases, uint64_t len, uint64_t cap_kmers,
                            uint64_t *skm_ret, uint32_t *skm_n, uint64_t *km_lo, uint64_t *km_hi, uint8_t *km_idx,
                            uint64_t *n_skm);
/* insert_superkmer: find-all then insert-missing for the k-mers of one vector, in order.
 * ids[i] = the entry's dense id, newly[i] = 1 if this call created it (its DATA is then
 * uninitialised, as in the reference).  HOST arrays; n <= 255. */
int brisk_hip_upsert_kmers(brisk_hip_index *h, const uint64_t *kmer_lo, const uint64_t *kmer_hi, const uint8_t *minimizer_idx,
                           uint64_t n, uint32_t *ids, uint8_t *newly);
/* get_superkmer / get: ids[i] = entry id or 0xffffffff when absent */
int brisk_hip_find_kmers(brisk_hip_index *h, const uint64_t *kmer_lo, const uint64_t *kmer_hi, const uint8_t *minimizer_idx,
                         uint64_t n, uint32_t *ids);
/* next(): as brisk_hip_enumerate, returning entry ids instead of counts */
int brisk_hip_enumerate_ids(brisk_hip_index *h, uint64_t *cursor, uint64_t *out_lo, uint64_t *out_hi,
                            uint8_t *out_minimizer_idx, uint32_t *out_ids, uint64_t cap, uint64_t *n_out);

/* ---- helpers on device buffers -------------------------------------------- */
int brisk_hip_pack_ascii(brisk_hip_index *h, const char *d_bases, uint64_t n_bases, uint32_t *d_packed);
/* synthetic reads of SURVEY.md 8(d), written packed; d_starts[n_reads+1] */
int brisk_hip_synth_reads(brisk_hip_index *h, uint64_t genome_len, uint64_t first_read, uint64_t n_reads,
                          uint32_t read_len, uint64_t seed_g, uint64_t seed_r,
                          uint32_t *d_packed, uint64_t *d_starts);

/* ---- test hooks -------------------------------------------------------------- */
/* order keys (bfc_hash_64, brisk/hashing.cpp:8-19) of n m-mers, computed by the device
 * code path the scan uses (table-driven class with guard band when exact == 0, the
 * plain FP64 fold when exact != 0).  HOST arrays. */
int brisk_hip_debug_order_keys(brisk_hip_index *h, const uint64_t *mmers, uint64_t n, int exact, uint64_t *keys);

/* ---- measurement ----------------------------------------------------------- */
/* With profiling on, every kernel launch is bracketed by HIP events on the
 * handle's stream.  brisk_hip_profile_read returns, per kernel slot, launches
 * and total milliseconds since the last reset. */
#define BRISK_HIP_PROFILE_SLOTS 16
int brisk_hip_profile_enable(brisk_hip_index *h, int on);
int brisk_hip_profile_read(brisk_hip_index *h, uint32_t *n_slots, const char **names, uint64_t *launches, double *ms);
int brisk_hip_profile_reset(brisk_hip_index *h);

#ifdef __cplusplus
}
#endif
#endif /* BRISK_HIP_H */

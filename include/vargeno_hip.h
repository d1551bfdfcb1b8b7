/* vargeno_hip.h -- C-ABI of the MI355X (gfx950) `vargeno geno` read-processing path.
 *
 * The reference has no plugin/FFI seam: the whole path is inlined in `static void genotype()`
 * (src/qv.cc:475-1787 of medvedevgroup/vargeno).  This header is the seam a maintainer would cut:
 * each entry point names the span of genotype() it replaces.  Plain pointers and sizes only; every
 * function returns 0 on success or a negative VG_E* code and never aborts (the reference asserts /
 * exit(EXIT_FAILURE)s instead: src/util.c:31-50, src/qv.cc:526-529).  One handle per GPU; a handle
 * is thread-compatible (one caller at a time).
 *
 * There is NO CPU fallback behind these calls: without a HIP device they fail with VG_ENODEV.
 */
#ifndef VARGENO_HIP_H
#define VARGENO_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define VG_OK        0
#define VG_EINVAL   -1   /* bad argument                                                     */
#define VG_EIO      -2   /* index file missing / short (reference: assert in util.c:31-50)   */
#define VG_ENOMEM   -3   /* host or device allocation failed                                 */
#define VG_ENODEV   -4   /* no HIP device / HIP runtime error (message via vg_last_error)    */
#define VG_ETOOBIG  -5   /* dictionary > 2^32 entries (reference: exit, qv.cc:526, 612)      */
#define VG_EBADREAD -6   /* a read longer than 1022 bases (reference BUF_SIZE, qv.cc:700)    */

typedef struct vg_index vg_index;        /* device-resident index + pile-up counters         */

/* The dictionaries and bit vectors exactly as the reference's files hold them, field by field
 * (dictgen.c:63-154 ref dict, :156-275 SNP dict; sdsl int_vector<1> payload words for the .bf
 * files).  Host pointers; copied, never retained. */
typedef struct {
	uint64_t n_ref;            const uint64_t *ref_kmer;  const uint32_t *ref_pos;  const uint8_t *ref_amb;
	uint64_t n_ref_aux;        const uint32_t *ref_aux;        /* [n_ref_aux][10]            */
	uint64_t n_snp;            const uint64_t *snp_kmer;  const uint32_t *snp_pos;
	const uint8_t *snp_info;   const uint8_t *snp_amb;    const uint8_t *snp_rf;    const uint8_t *snp_af;
	uint64_t n_snp_aux;        const uint32_t *snp_aux_pos;    /* [n_snp_aux][10]            */
	                           const uint8_t  *snp_aux_info;   /* [n_snp_aux][10]            */
	uint64_t ref_bf_bits;      const uint64_t *ref_bf_words;   /* >= ceil(min(bits,2^32)/64) */
	uint64_t snp_bf_bits;      const uint64_t *snp_bf_words;   /* >= ceil(bits/64)           */
} vg_index_arrays;

/* Counters mirroring the reference's `#if DEBUG` set (qv.cc:737-751) plus the event counts that
 * price algorithmic bytes.  All accumulate since open / vg_counts_reset. */
typedef struct {
	uint64_t reads, reads_n, reads_invalid, passes, passes_ok, chunks, gate_open,
	         refbf_pos, snpbf_pos, large_block, ref_query, snp_query, ref_probe, snp_probe,
	         scan_ref, scan_snp, scan_oob, aux_ref, aux_snp, site_test, ctx, walks, incr,
	         ingest_bytes;
	uint64_t overflow_reads;     /* reads that outgrew the main wave tier's LDS tables and were redone by
	                                the deep wave tier (same kernel, tables 3-7x deeper)       */
	uint64_t overflow_deep;      /* of those, reads that outgrew the deep lists too and went to the
	                                generic lane tier (one read per lane, lists in HBM scratch)   */
	uint64_t alg_bytes;          /* sum of unit cost x event count                            */
} vg_stats;

/* Kernel timing, averaged over the batches processed since the previous vg_timing_get, from HIP
 * events recorded on the handle's own streams. */
typedef struct {
	float ms_total;              /* first kernel start -> last kernel end of a batch           */
	float ms_pack;               /* vg_pack_kernel: ASCII -> 2-bit chunk k-mers + gate bits    */
	float ms_main;               /* vg_wave_kernel, the dominant kernel                        */
	float ms_tail;               /* the spill tiers (deep-list wave tier + generic lane tier) for the
	                                reads that outgrew the LDS lists: second stream, under the next
	                                batches' kernels                                             */
	uint32_t batches;
	float ms_deep_lists;         /* of ms_tail: the deep-list wave tier                        */
} vg_timing;

const char *vg_last_error(void);
/* sha256 prefix of the sources this library was compiled from (csrc/Makefile): lets a caller refuse a stale build */
const char *vg_build_id(void);
int  vg_device_count(void);
/* Total memory of a device in bytes (0 on failure): what a caller that puts SEVERAL replicas on one device divides into the
 * budgets it hands to vg_index_open_ex -- without a budget every replica plans for the whole device (less 12 GiB), and the
 * third or fourth one fails with VG_ENOMEM.  vg_share_budget() is that division as the CLI makes it:
 * (min(total, free at the time of the call) - 12 GiB) / replicas_on_the_device.  "Free at the time of the call" makes it ONE
 * number for all sharers only when it is asked for before any of them has opened: a process that opens its replicas itself calls
 * it once per device, before the first vg_index_open_ex, and hands every sharer that number (the command line; it also does so
 * for a single replica when it has taken a read store on the device).  Sharers in different processes must agree on a number
 * among themselves (bench.py: every rank calls before any opens, the minimum over the ranks is everybody's budget) -- a rank that
 * calls after a sibling has taken its block would get half a share. */
uint64_t vg_device_memory(int device);
uint64_t vg_share_budget(int device, int replicas_on_the_device);
/* Host -> device rate of page-locked memory over this device's link in bytes per second, measured now (three 64 MiB copies; 0 on
 * failure): what FASTQ text framed ON the device can arrive at -- the command line compares it with the rate its host threads
 * frame + pack at (vg_packer_*) and lets the faster route take the file (r05; r04 chose by the number of CPUs). */
double vg_link_rate(int device);

/* Page-locked host buffers for the batches / FASTQ chunks handed to vg_reads_submit / vg_fastq_submit
 * (optional: any host memory works, pinned memory copies at link speed).  NULL on failure. */
void *vg_host_alloc_pinned(size_t bytes);
void  vg_host_free_pinned(void *p);

/* Replaces the loader half of genotype(): qv.cc:519-695 (dict files -> jump tables, dict arrays,
 * aux tables, pile-up seeding) and main()'s BloomFilter::load (qv.cc:2140-2144).
 * vg_index_open reads <prefix>.ref.dict/.snp.dict/.ref.bf/.snp.bf; vg_index_create takes the same
 * content from memory. */
int  vg_index_open(const char *prefix, int device, vg_index **out);
/* The same with a device-memory budget for this replica (bytes; 0 = the device's whole memory less 12 GiB, which is what
 * vg_index_open assumes -- $VG_MAX_DEVICE_BYTES, when set, is its budget).  Which optional views are built is decided from the
 * dictionaries' sizes and the budget alone, in a fixed order, before anything is allocated -- never from what happens to be free at
 * that moment: the same files and the same budget always give the same vg_index_views().  A caller that shares the device must say
 * how much of it is its own; when an allocation the plan had room for fails all the same, the call fails with VG_ENOMEM instead of
 * silently building a slower layout.  VG_ENOMEM also when the budget is below the smallest layout.
 * The budget bounds what the FINISHED handle holds (vg_index_device_bytes <= budget; profiles/budget_sweep_r05.json).  With
 * every view planned, construction stays inside the handle's one block (vg_arena.h: the order of construction keeps the live set
 * within the finished size).  With views left out the finished handle is smaller than what construction has alive at its peak --
 * the dictionaries' columns beside their entries, the sorts' buffers -- and those temporaries are taken from the device beside the
 * block and given back: up to ~100 GB more than the budget for some hundred milliseconds at hg38 scale.  Replicas that SHARE a
 * device and open at the same time must leave that room (or open one after the other).  When the device does not have that room
 * free at the time of the call (a plan limited by the device's own size) the handle is built without the block, one allocation
 * per buffer: slower to open, same views, same results (r06).
 * Tables scale with the index (r06): the reference's jump table and the direct table have ~2 entries per k-mer instead of 2^32
 * (a chr22-scale index holds ~8 GB instead of 92); vg_index_plan() names the widths, and under a budget the direct table is
 * tried with half and a quarter of its buckets before it is given up. */
int  vg_index_open_ex(const char *prefix, int device, uint64_t max_device_bytes, vg_index **out);
/* What the budget bought, in words: planned bytes, views kept, views left out with what each costs ("" for a null handle).
 * The string lives as long as the handle. */
const char *vg_index_plan(const vg_index *ix);
/* Where the handle's start-up time went (the reference's loader, qv.cc:519-695, takes ~240 s at hg38 scale):
 * wall seconds of each phase of vg_index_open / vg_index_create with the part spent inside allocation calls, then the memory the
 * handle ended up with (r05: one block taken from the driver once and carved up, vg_arena.h).  "" for a null handle;
 * the string lives as long as the handle. */
const char *vg_index_open_report(const vg_index *ix);
int  vg_index_create(const vg_index_arrays *a, int device, vg_index **out);
void vg_index_close(vg_index *ix);               /* qv.cc:1775-1786 */

/* Device bytes held by the index (tables + pile-up + scratch). */
uint64_t vg_index_device_bytes(const vg_index *ix);

/* Which optional re-laid-out views of the dictionaries the handle holds (bit mask).  They change speed, never results: a
 * view is left out when the index is too large for it or the device had too little free memory while the handle was
 * built (or a VG_NO_* development switch said so) -- a caller that cares about throughput should look. */
#define VG_VIEW_SEC        1u   /* LO32-ordered view of the reference dictionary (high-half neighbours, qv.cc:1213-1296)   */
#define VG_VIEW_MX         2u   /* merged exact-match view of both dictionaries (qv.cc:840-841)                            */
#define VG_VIEW_DX         4u   /* direct table over HI32 in front of the merged view                                      */
#define VG_VIEW_SNP_PROBE  8u   /* strided-probe view of the SNP dictionary (iterate_snp_dict, qv.cc:413-464)              */
#define VG_VIEW_SNP_JG32  16u   /* HI32 jump table of the SNP dictionary (indexes too large for the merged view)           */
#define VG_VIEW_SNP_SIG   64u   /* ... its 16-bit signature form (the default; the probe view is built under VG_NO_SIG_VIEW)    */
#define VG_VIEW_SEC_IS_BF 128u  /* the reference bit vector was verified to be the LO32 set of the dictionary: the LO32-ordered
                                   view answers its probes too (false e.g. for an index built from a soft-masked FASTA)   */
#define VG_VIEW_HX        32u   /* paired HI32 table of both dictionaries (indexes too large for the merged view)          */
#define VG_VIEW_SSEC     256u   /* LO32-ordered view of the SNP dictionary (high-half SNP neighbours, qv.cc:1303-1352; r06) */
uint32_t vg_index_views(const vg_index *ix);

/* Replaces the FASTQ loop body, qv.cc:760-1558, for a batch of reads: flat ASCII bases and quality
 * characters (same offsets; offsets[n_reads] = total length) in HOST memory.  Copies to the
 * device and enqueues the kernels on the handle's stream; returns before they finish. */
int  vg_reads_submit(vg_index *ix, const uint8_t *bases, const uint8_t *quals,
                     const uint64_t *offsets, uint64_t n_reads);

/* Same, for a batch already resident in device memory (hipMalloc'd by the caller on ix's device):
 * what bench.py times.  The handle works on its own non-blocking streams: the buffers must be complete when the
 * call is made (synchronise the stream that produced them) and stay valid and unchanged until the next vg_sync. */
int  vg_reads_process_device(vg_index *ix, const uint8_t *d_bases, const uint8_t *d_quals,
                             const uint64_t *d_offsets, uint64_t n_reads);

/* The same batch with the quality strings reduced to what the path reads of them: ONE GATE WORD per read, bit c set iff quality
 * character c of the read is below '8' -- the only question qv.cc:836, 943 asks, with c the CHUNK number (a read of at most
 * 1022 bases has at most 31 chunks).  The reference's loop never looks at the other characters; handing the strings over makes
 * the device fetch every line of them for four characters per 150 bp read, as much traffic again as the bases.  This is the form
 * the device-side FASTQ framing (vg_fastq_stream_push) produces for itself, and what bench.py's resident batches use.
 * A read of more than 32 chunks (1055 bases: no FASTQ line the reference can read holds one) is counted in reads_invalid. */
int  vg_reads_process_device_gated(vg_index *ix, const uint8_t *d_bases, const uint32_t *d_gate_words,
                                   const uint64_t *d_offsets, uint64_t n_reads);

/* A batch that is already framed and 2-bit packed, in HOST memory -- SURVEY.md §8b's "pre-packed 2-bit + the <= 31 quality chars the
 * gate can see", the latter reduced to the comparison's result: kmers = the reads' chunk k-mers one read after the other
 * (encode_kmer, util.c:89-111: A0 C1 G2 T3, character k[0] in bits 0-1), chunk_offsets[r] = chunks before read r
 * (chunk_offsets[0] = 0, [n_reads] = total), meta[r] = gate bits of read r (bit c = quality character c < '8', qv.cc:836, 943) with
 * bit 62 set for a read the reference skips (an N in its trimmed part, qv.cc:815-828) and bit 63 for one it aborts on (another
 * character outside ACGTacgt, util.c:103) -- 48 bytes per 150 bp read where its FASTQ record has ~315.  Blocking copies: the
 * arrays are free when the call returns; the read loop is only enqueued.  VG_EBADREAD: a read of more than 31 chunks. */
int  vg_reads_submit_packed(vg_index *ix, const uint64_t *kmers, const uint64_t *meta, const uint64_t *chunk_offsets, uint64_t n_reads);
/* The same for arrays in PAGE-LOCKED memory (vg_host_alloc_pinned) that the caller leaves untouched until vg_sync: the copies are
 * asynchronous and the call returns as soon as the batch is enqueued (r05: the command line hands over the hundreds of batches it
 * packed while the index was being opened). */
int  vg_reads_submit_packed_async(vg_index *ix, const uint64_t *kmers, const uint64_t *meta, const uint64_t *chunk_offsets, uint64_t n_reads);

/* A read store: packed batches parked in DEVICE memory before an index handle exists on that device -- a job's FASTQ file is framed
 * and packed (vg_packer_*) while vg_index_open runs, and what is packed goes up the otherwise idle link at once instead of waiting
 * in page-locked host memory (which costs ~0.15 s per GB to lock and ~0.1 s per GB to give back at exit).  The reference has no
 * counterpart: it reads the file after load_dict_from_file has returned (qv.cc:1757-1767, then 760-784).
 *   vg_read_store_create   takes max_bytes of device memory at once (before the index is planned, so that the plan sees what is left)
 *   vg_read_store_push     validates one batch (the rules of vg_reads_submit_packed) and enqueues its copies; the arrays -- page-locked
 *                          memory copies at link speed -- must stay untouched until the NEXT vg_read_store_push / _flush on this
 *                          store returns (a caller alternates between two sets).  VG_ENOMEM: the store is full; the batch was not
 *                          taken and the caller sends it, and what follows, down another route
 *   vg_reads_submit_store  every batch of the store through the read loop of an open handle on the same device, in push order,
 *                          asynchronously (vg_sync); the store must live until then.  A store may be submitted more than once
 *   vg_read_store_destroy  gives the memory back */
typedef struct vg_read_store vg_read_store;
int      vg_read_store_create(int device, uint64_t max_bytes, vg_read_store **out);
int      vg_read_store_push(vg_read_store *rs, const uint64_t *kmers, const uint64_t *meta, const uint64_t *chunk_offsets, uint64_t n_reads);
int      vg_read_store_flush(vg_read_store *rs);
uint64_t vg_read_store_reads(const vg_read_store *rs);
uint64_t vg_read_store_bytes_used(const vg_read_store *rs);
int      vg_reads_submit_store(vg_index *ix, vg_read_store *rs);
void     vg_read_store_destroy(vg_read_store *rs);

/* Same again, starting from raw FASTQ text (host memory): replaces the four fgets() + strlen of qv.cc:760-784.
 *
 * Stream form -- the caller only moves bytes.  vg_fastq_stream_push takes the next chunk of the file, cut anywhere; the device
 * frames the complete 4-line records (carrying the unfinished last record of a chunk over to the next one by itself) and runs
 * them through the read loop.  The call returns when the chunk has been copied to the device (its buffer is free again; pinned
 * memory copies at link speed); framing and processing are only enqueued, and nothing is reported back until
 * vg_fastq_stream_end: records framed, bytes consumed (= offset of the first byte the device did not process: the incomplete tail
 * of the file, or everything from the first chunk it refused), offset of the last framed record (a host reader that must
 * reproduce the reference's stale-buffer behaviour on a truncated final record starts there), and whether a chunk was refused.
 * A chunk is refused -- and with it everything after it -- when it holds a line longer than fgets' 1023 characters, a quality
 * line shorter than the read's chunk count, or lines shorter than 8 bytes on average: frame the rest on the host
 * (vg_reads_submit).  One stream at a time per handle; chunks of less than 2 GiB. */
int  vg_fastq_stream_begin(vg_index *ix);
/* The same stream with the framing AND the 2-bit packing done by host threads inside the library (r04): the rules
 * are the device framing's (records = four lines counted from the start of the stream, lines of at most 1023 characters, a
 * quality character for every chunk; anything else refuses the chunk and the rest of the stream), but what crosses the link is
 * the packed form (8 bytes per chunk + 16 per read instead of the text: 6.5 x fewer bytes for 150 bp reads), so a host with cores
 * to spare ingests several times faster than the link can move text.  host_threads > 0: that many; 0: device framing after all
 * (same as vg_fastq_stream_begin); < 0: the library decides (half the CPUs the process may use -- a cgroup CPU quota counts --,
 * at most 96; device framing when that is fewer than 32; $VG_PACK_THREADS overrides).  At most 256 threads are started.  push / end are the calls above and report the same things. */
int  vg_fastq_stream_begin_packed(vg_index *ix, int host_threads);
int  vg_fastq_stream_push(vg_index *ix, const uint8_t *text, uint64_t nbytes);
int  vg_fastq_stream_end(vg_index *ix, uint64_t *n_records, uint64_t *consumed, uint64_t *last_record_start, int *refused);

/* One self-contained chunk, synchronously: *consumed = bytes used (the rest, an incomplete last record, is the caller's to
 * resubmit with the next chunk); waits for the framing, not for the read loop.  VG_EBADREAD: the chunk was refused (see above);
 * nothing was processed. */
int  vg_fastq_submit(vg_index *ix, const uint8_t *text, uint64_t nbytes,
                     uint64_t *n_records, uint64_t *consumed, uint64_t *last_record_start);

/* The host-side framing + packing on its own (no device is touched): for a caller that packs on its own threads and hands the
 * batches to vg_reads_submit_packed.  A packer is a stream like the one above: push chunks cut anywhere; each push returns the
 * complete records it framed (arrays sized with vg_packer_reads_cap / _kmers_cap of the chunk's length); vg_packer_end reports
 * records, bytes consumed, the start of the last framed record and whether a chunk was refused (nothing of a refused chunk or of
 * what follows it is returned). */
typedef struct vg_packer vg_packer;
int  vg_packer_create(int host_threads, vg_packer **out);
void vg_packer_destroy(vg_packer *pk);
int  vg_packer_begin(vg_packer *pk);
uint64_t vg_packer_reads_cap(uint64_t nbytes);
uint64_t vg_packer_kmers_cap(uint64_t nbytes);
int  vg_packer_push(vg_packer *pk, const uint8_t *text, uint64_t nbytes, uint64_t *kmers, uint64_t kmers_cap, uint64_t *meta,
                    uint64_t *chunk_offsets, uint64_t reads_cap, uint64_t *n_reads, uint64_t *n_chunks, uint64_t *n_invalid);
int  vg_packer_end(vg_packer *pk, uint64_t *n_records, uint64_t *consumed, uint64_t *last_record_start, int *refused);

int  vg_sync(vg_index *ix);                       /* drain the stream                          */
int  vg_stats_get(vg_index *ix, vg_stats *out);   /* implies vg_sync                           */
int  vg_set_stats(vg_index *ix, int enable);      /* event counting on (default) / off: the
                                                     counting build of the kernel carries ~50 extra
                                                     registers per lane, so timed runs switch it off */
int  vg_timing_get(vg_index *ix, vg_timing *out);

/* SNP sites = positions seeded with ref != alt (qv.cc:637-659, 1580), ascending.
 * Counters are exact sums; the reference's 6-bit saturation (vartype.h:27, qv.cc:1411, 1419) is
 * min(63, sum) and is applied by vg_counts_fetch. */
uint64_t vg_num_sites(const vg_index *ix);
int  vg_sites_fetch(vg_index *ix, uint32_t *pos, uint8_t *ref_base, uint8_t *alt_base,
                    uint8_t *ref_freq, uint8_t *alt_freq);
int  vg_counts_fetch(vg_index *ix, uint8_t *ref_cnt, uint8_t *alt_cnt);    /* clamped at 63     */
int  vg_counts_reset(vg_index *ix);

/* Device pointer to the raw u32 counter array, length 2 * vg_num_sites: [2*s] = ref, [2*s+1] = alt
 * (the call first drains the batches in flight).
 * This is the ONE buffer that crosses GPUs: sum it over ranks (RCCL all-reduce over xGMI) before
 * vg_counts_fetch.  Reads shard with no other exchange. */
int  vg_counts_device_ptr(vg_index *ix, void **d_counts, uint64_t *n_u32);

/* All-reduce (sum) the counters over the ranks of an RCCL communicator (ncclComm_t passed as
 * void*), on the handle's stream. */
int  vg_counts_allreduce(vg_index *ix, void *nccl_comm);

/* The same exchange for ONE process that drives n devices (one handle each, all replicas of one index): builds an RCCL
 * communicator over the handles' devices, all-reduces every replica's counters in place, tears the communicator down.
 * This is what `vargeno geno` calls with VARGENO_GPUS=n.  n = 1 is the identity (and still goes through RCCL).
 * Replicas that share a device are summed on that device first; one of them joins the all-reduce, all get the result. */
int  vg_counts_allreduce_devices(vg_index **handles, int n);

#ifdef __cplusplus
}
#endif
#endif

/*
 * libsnappyhash -- C ABI of the MI355X (gfx950) asset hot path.
 *
 * Every entry point replaces one managed call of the reference plugin (paths
 * are relative to /root/reference).  The reference has no native interface for
 * this path: the seam is the static method `string Crypto.GetHash(byte[])`
 * (Snappy/Utils/Crypto.cs:8) with its three call sites, and the idiom
 * `new LZ4Stream(fs, Decompress, ...).CopyTo(ms)` (Snappy/Features/Mcdf/
 * McdfManager.cs:73-79).  The shapes below are what a P/Invoke shim for those
 * call sites binds.
 *
 * Conventions
 *   - Plain pointers and sizes only.  The caller owns every buffer; the library
 *     keeps no pointer after a call returns and hands out no memory.
 *   - Every call is synchronous and blocking, like the managed originals, and
 *     safe to call from several threads at once (the reference hashes from the
 *     UI thread and from a thread-pool task: Snappy/Plugin.cs:257-278).
 *   - Return value: SNAPPY_OK (0) or a negative SNAPPY_E_* code.  Nothing throws
 *     or aborts across the ABI.  There is no CPU fallback: without a usable
 *     gfx950 device every compute entry point returns SNAPPY_E_NO_DEVICE.
 *   - "host" pointers are ordinary process memory.  "device" pointers are HIP
 *     device allocations on the device named by the call.
 */
#ifndef SNAPPYHASH_H
#define SNAPPYHASH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status */
#define SNAPPY_OK                  0
#define SNAPPY_E_INVALID_ARG     (-1)  /* null pointer, bad device index, range overflow          */
#define SNAPPY_E_NO_DEVICE       (-2)  /* no gfx950 device / HIP runtime unusable                 */
#define SNAPPY_E_HIP             (-3)  /* a HIP call failed; see snappy_last_error_detail()       */
#define SNAPPY_E_NOMEM           (-4)  /* host or device allocation failed                        */
/* lz4net LZ4Stream read errors (the managed reader throws in each of these cases) */
#define SNAPPY_E_LZ4_TRUNCATED   (-10) /* EOF inside a varint or a chunk payload (EndOfStream)    */
#define SNAPPY_E_LZ4_BAD_LENGTHS (-11) /* compressedLength > originalLength, or negative as int   */
#define SNAPPY_E_LZ4_PASSES      (-12) /* chunk flags ask for multi-pass decode (NotSupported)    */
#define SNAPPY_E_LZ4_CORRUPT     (-13) /* block does not decode to exactly originalLength bytes   */
#define SNAPPY_E_CAPACITY        (-14) /* caller's output buffer is too small                     */
/* McdfHeader.ReadHeader / FromBinaryReader errors (Snappy/Features/Mcdf/McdfHeader.cs:26-62) */
#define SNAPPY_E_MCDF_MAGIC      (-20) /* first four bytes are not "MCDF"                         */
#define SNAPPY_E_MCDF_VERSION    (-21) /* version byte != 1                                       */
#define SNAPPY_E_MCDF_TRUNCATED  (-22) /* payload ends inside the header                          */
#define SNAPPY_E_MCDF_JSON       (-23) /* header JSON does not parse into McdfData                */
#define SNAPPY_E_MCDF_BASE64     (-24) /* CustomizePlusData is not Base64 (FormatException,       */
                                       /* Convert.FromBase64String at McdfManager.cs:136)        */
/* UpdateSnapshot: an existing snapshot.json / *_history.json does not deserialize
 * (JsonConvert.DeserializeObject throws, Snappy/Core/SnapshotManager.cs:434-450) */
#define SNAPPY_E_SNAPSHOT_JSON   (-30)

/* Per-file status written by the MCDF entry points (McdfManager.cs:36-63). */
#define SNAPPY_FILE_HASHED             0
#define SNAPPY_FILE_SKIPPED_EMPTY      1 /* Length == 0: `continue` at McdfManager.cs:39-40       */
#define SNAPPY_FILE_SKIPPED_SHORT_READ 2 /* ReadBytes came up short: logged + skipped, :43-49     */

/* --------------------------------------------------------------- lifecycle */

/* Binds the first n_devices visible gfx950 devices (n_devices <= 0: all).
 * Optional: every entry point initialises on first use with all devices.
 * Replaces nothing in the reference (managed code needs no set-up). */
int snappy_init(int n_devices);
/* Releases streams, staging buffers and device arenas.  Not safe while other
 * calls are in flight. */
void snappy_shutdown(void);
/* Number of devices the library is using (0 before a successful init). */
int snappy_device_count(void);
/* Static, NUL-terminated description of a status code. */
const char *snappy_strerror(int code);
/* Thread-local detail of the last SNAPPY_E_HIP on this thread ("" if none). */
const char *snappy_last_error_detail(void);

/* Page-locked host memory for file contents.  The reference reads every file
 * into a managed array (File.ReadAllBytes, Snappy/Core/SnapshotManager.cs:317,
 * Snappy/Data/SnapshotMigrator.cs:73); a caller that reads into a buffer from
 * snappy_host_alloc instead lets the entry points below move it to HBM by
 * direct DMA at PCIe rate, without the staging copy that ordinary (pageable)
 * memory needs.  Any host pointer is accepted everywhere; this is only faster.
 * Free with snappy_host_free.  No counterpart in the reference. */
int snappy_host_alloc(uint64_t bytes, void **out);
int snappy_host_free(void *ptr);

/* ------------------------------------------------------------------- SHA-1 */

/* Crypto.GetHash without the formatting (Snappy/Utils/Crypto.cs:10-11,
 * SHA1.Create().ComputeHash(bytes)): 20-byte SHA-1 of data[0..len).
 * data may be NULL when len == 0. */
int snappy_sha1(const uint8_t *data, uint64_t len, uint8_t out20[20]);

/* Exactly Crypto.GetHash (Snappy/Utils/Crypto.cs:8-22): 40 UPPERCASE hex
 * characters followed by NUL. */
int snappy_sha1_hex(const uint8_t *data, uint64_t len, char out41[41]);

/* The three hash loops collapsed (Snappy/Core/SnapshotManager.cs:312-324,
 * Snappy/Features/Mcdf/McdfManager.cs:36-63, Snappy/Data/SnapshotMigrator.cs:
 * 62-86): digests[20*i .. 20*i+20) = SHA-1(base + offsets[i], lengths[i]).
 * All pointers are host memory.  lengths[i] == 0 is allowed (digest of the
 * empty message); the MCDF loop's skip of empty entries is the caller's. */
int snappy_sha1_batch(const uint8_t *base, const uint64_t *offsets,
                      const uint64_t *lengths, uint64_t n, uint8_t *digests);

/* Same loop for files read into separate arrays (File.ReadAllBytes per file,
 * SnapshotManager.cs:317): digests[20*i..] = SHA-1(datas[i], lengths[i]). */
int snappy_sha1_files(const uint8_t *const *datas, const uint64_t *lengths,
                      uint64_t n, uint8_t *digests);

/* Crypto.ToHashedString for n digests (Snappy/Utils/Crypto.cs:14-22):
 * hex[41*i..] = 40 uppercase hex chars + NUL.  Pure host formatting. */
int snappy_digests_to_hex(const uint8_t *digests, uint64_t n, char *hex);

/* snappy_sha1_batch over data that already lives in HBM on `device`
 * (0 <= device < snappy_device_count()).  d_base is a device pointer; offsets,
 * lengths and digests are host memory.  `stream` is a hipStream_t on that
 * device or NULL for a library stream; the call returns after the digests
 * have landed.  Any byte offset works; files that start on 64-byte boundaries
 * hash a few percent faster when there are many of them.  No counterpart in the reference (it hashes managed arrays);
 * it is the loop above for callers that keep assets resident on the GPU. */
int snappy_sha1_batch_device(int device, const void *d_base, const uint64_t *offsets,
                             const uint64_t *lengths, uint64_t n, uint8_t *digests,
                             void *stream);

/* The same with everything resident: d_offsets, d_lengths (n entries each) and
 * d_digests (20*n bytes) are device memory too, so nothing but the launch
 * crosses PCIe; the size ordering and the cut into execution paths are computed on
 * the device.  Returns after the digests are in d_digests.  No counterpart in
 * the reference; for callers that keep asset tables on the GPU. */
int snappy_sha1_batch_resident(int device, const void *d_base, const uint64_t *d_offsets,
                               const uint64_t *d_lengths, uint64_t n, uint8_t *d_digests,
                               void *stream);

/* snappy_sha1_batch_resident for callers that cannot allocate device memory:
 * code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  Same arguments, same digests; no counterpart in the reference.
 * The planner's header and the sorted task list live in a fixed array inside
 * the library (24 MiB + 16 KiB per GPU, held until the process ends) instead of
 * pooled buffers, so the call issues no hipMalloc / hipFree / stream or event
 * creation of its own.
 * ENQUEUE ONLY: unlike snappy_sha1_batch_resident the call does not wait.  It puts
 * one memset and four kernel launches on `stream` and returns; the digests are
 * valid once the stream has reached that point (hipStreamSynchronize, an event, or
 * the replay of a graph the call was captured into).  It may be captured: it
 * neither synchronises nor allocates.  d_offsets / d_lengths / d_digests are read
 * and written by the GPU when the work runs, not during the call.
 * Limits: at most SNAPPY_NOALLOC_MAX_FILES files per call (SNAPPY_E_CAPACITY
 * beyond); snappy_init must have run (the call does not initialise the
 * library); `stream` NULL means the device's default stream, which cannot be
 * captured.  All calls on one GPU share the array and nothing in it survives
 * from one call to the next: issue them on ONE stream (or order them with
 * events); work of two calls that overlaps on a GPU races. */
#define SNAPPY_NOALLOC_MAX_FILES (1u << 20)
int snappy_sha1_batch_resident_noalloc(int device, const void *d_base,
                                       const uint64_t *d_offsets, const uint64_t *d_lengths,
                                       uint64_t n, uint8_t *d_digests, void *stream);

/* How snappy_sha1_batch / snappy_sha1_files deal a batch out to the devices
 * (files are independent, nothing is exchanged): longest processing time first
 * by padded 64-byte blocks, (len + 8) / 64 + 1 per file.  Files of at least 1/64
 * of a device's average share are sorted by size and each goes to the device
 * with the fewest blocks so far; the small files that remain fill the devices up
 * to the average in caller order, as contiguous runs.  shard_of[i] receives the
 * shard of file i; shard_blocks (may be NULL) the n_shards block totals.  Host
 * only; exposed so that a caller running one process per GPU shards a file list
 * the same way.  No counterpart in the reference (its loop is sequential,
 * Snappy/Core/SnapshotManager.cs:312-324). */
int snappy_plan_shards_lpt(const uint64_t *lengths, uint64_t n, uint32_t n_shards,
                           uint32_t *shard_of, uint64_t *shard_blocks);

/* The simpler split into contiguous groups of near-equal byte count: shard k is
 * cuts[k] .. cuts[k+1] (cuts has n_shards + 1 entries).  For callers whose
 * shards must be ranges of the list.  Host only; no counterpart in the reference. */
int snappy_plan_shards(const uint64_t *lengths, uint64_t n, uint32_t n_shards, uint64_t *cuts);

/* Diagnostic: what the host-pointer entry points have given device `device`
 * since it was bound -- files (archives for the MCDF entry points) and their
 * bytes, counted where the device's slot is held.  Shows where a sharded call
 * actually ran.  No counterpart in the reference. */
int snappy_device_work(int device, uint64_t *units, uint64_t *bytes);

/* ----------------------------------------------------- lz4net stream + MCDF */

/* Walks the chunk headers of an lz4net LZ4Stream (what McdfManager.cs:73-79
 * reads) and returns the decoded size: the length the MemoryStream would have
 * after CopyTo.  Host only.  Reports the structural errors
 * (TRUNCATED / BAD_LENGTHS / PASSES), not block corruption. */
int snappy_lz4net_decoded_size(const uint8_t *in, uint64_t in_len, uint64_t *out_len);

/* `lz4Stream.CopyTo(memoryStream)` (McdfManager.cs:72-80): decodes every chunk
 * of the stream into out[0..*out_len).  Host pointers; the LZ4 blocks are
 * decoded on the GPU. */
int snappy_lz4net_decode(const uint8_t *in, uint64_t in_len, uint8_t *out,
                         uint64_t out_cap, uint64_t *out_len);

/* McdfHeader.FromBinaryReader (Snappy/Features/Mcdf/McdfHeader.cs:26-62) on a
 * decoded payload: checks "MCDF" + version 1, parses the JSON (McdfData.cs:
 * 9-13,25) and reports the file table.  Host only.
 *   *n_files      number of entries of Files
 *   *body_offset  offset of the first file body in the payload
 * file_lengths may be NULL to query *n_files; otherwise it receives up to
 * lengths_cap entries (SNAPPY_E_CAPACITY when too small). */
int snappy_mcdf_parse_header(const uint8_t *payload, uint64_t payload_len,
                             uint64_t *n_files, uint64_t *body_offset,
                             int64_t *file_lengths, uint64_t lengths_cap);

/* ImportMcdf data path, fused (McdfManager.cs:72-80 decode + :28-66 slice and
 * hash): decodes the lz4net stream on the GPU, parses the header on the host,
 * and hashes every file body where it lies in HBM.
 *   digests      n_files*20 bytes; entries whose status is not HASHED are zero
 *   statuses     n_files bytes of SNAPPY_FILE_*
 *   out_payload  NULL, or room for the decoded payload (out_cap bytes)
 * n_files_cap is the capacity of digests/statuses in entries. */
int snappy_mcdf_decode_and_hash(const uint8_t *in, uint64_t in_len,
                                uint8_t *digests, uint8_t *statuses, uint64_t n_files_cap,
                                uint64_t *n_files, uint64_t *body_offset,
                                uint8_t *out_payload, uint64_t out_cap, uint64_t *payload_len);

/* The same import for several archives at once (the loop a bulk import would
 * run over McdfManager.ImportMcdf, McdfManager.cs:68): one decode launch and one
 * hash launch cover every archive, which is what fills the GPU -- an archive is
 * only a few hundred 1 MiB chunks.  digests/statuses hold the Files entries of
 * archive 0, then archive 1, ... (n_files[a] entries each; n_files_cap is their
 * total capacity).  archive_status[a] is that archive's SNAPPY_* status; an
 * archive that failed contributes no entries and does not disturb the others
 * (one whose declared size cannot be allocated next to the rest gets
 * SNAPPY_E_NOMEM).  With several devices the archives are sharded by archive,
 * largest decoded size first onto the least loaded device, so that each payload
 * is decoded and hashed in one device's HBM.  body_offsets and payload_lens may
 * be NULL. */
int snappy_mcdf_decode_and_hash_batch(uint64_t n_archives, const uint8_t *const *ins,
                                      const uint64_t *in_lens, uint8_t *digests,
                                      uint8_t *statuses, uint64_t n_files_cap,
                                      uint64_t *n_files, uint64_t *body_offsets,
                                      uint64_t *payload_lens, int *archive_status);

/* ------------------------------------------------ headless callers (files) */

/* The hash-and-store loop with its file system side: for every path that names
 * an existing file, read it whole (File.ReadAllBytes, SnapshotManager.cs:317),
 * hash it (all files in one GPU batch), write hex[41*i..] = its 40 uppercase hex
 * characters + NUL, and -- when files_dir is not NULL -- store the bytes as
 * files_dir/<HASH>.dat unless that file exists (SnapshotManager.cs:453-475; the
 * migrator's variant is Snappy/Data/SnapshotMigrator.cs:62-86).
 * status[i]: 0 = hashed, 1 = no such file (the loops `continue`, hex[41*i] = NUL). */
int snappy_store_files(const char *const *paths, uint64_t n, const char *files_dir,
                       char *hex, uint8_t *status);

/* SnapshotManager.UpdateSnapshot for the local player without the game and the
 * plugin IPC: what a press of "save snapshot" leaves on disk.
 * Inputs are what Penumbra / Glamourer / Customize+ hand the plugin
 * (Snappy/Core/SnapshotManager.cs:295-307): n_files resolved file paths, file i
 * with paths_per_file[i] game paths (all files' game paths one after the other
 * in game_paths), and the three state strings (NULL = empty).
 *   - every resolved path that names an existing file is read and hashed (one GPU
 *     batch); resolvedPaths[hash] = path, last writer wins;
 *     newFileReplacements[gamePath] = hash, case-sensitive (:312-324);
 *   - snapshot_dir and snapshot_dir/_files are created; an existing
 *     snapshot.json, glamourer_history.json and customize_history.json are read
 *     back (:424-451; SNAPPY_E_SNAPSHOT_JSON when one does not parse);
 *   - the new pairs are merged into FileReplacements (OrdinalIgnoreCase keys:
 *     the first spelling stays, the last hash wins) and _files/<HASH>.dat is
 *     written unless present (:453-475).  The bytes written are the ones that
 *     were hashed, so no source is read twice and a hash shared by many game
 *     paths is stored once;
 *   - ManipulationString is replaced, a Glamourer / Customize+ history entry is
 *     appended when the value differs from the last entry's (:477-511; the
 *     Customize+ data is stored as Base64 of its UTF-8 bytes, its importable
 *     template left empty), LastUpdate is set, and the three files are written
 *     as JsonConvert does with Formatting.Indented (:672-695).
 * source_actor names a new snapshot (SnapshotInfo.SourceActor); an existing one
 * keeps its own.  hex (41*n_files bytes) and status (n_files bytes) may be NULL;
 * otherwise as snappy_store_files. */
int snappy_update_snapshot(const char *snapshot_dir, const char *source_actor,
                           const char *const *resolved_paths, const char *const *game_paths,
                           const uint32_t *paths_per_file, uint64_t n_files,
                           const char *glamourer, const char *customize,
                           const char *manipulation, char *hex, uint8_t *status);

/* McdfManager.ImportMcdf without UI and plugin IPC (McdfManager.cs:68-156):
 * reads the archive, decodes and hashes it on the GPU, and writes the snapshot
 * folder working_dir/<Description or MCDF_Import_yyyyMMddHHmmss> (replaced when
 * it exists): _files/<HASH>.dat for every file body, snapshot.json (Snappy/
 * Models/SnapshotInfo.cs:6-14), glamourer_history.json, customize_history.json
 * (SnapshotManager.cs:672-695; the importable Customize+ template is left
 * empty).  snapshot_dir_out (may be NULL) receives the folder's path.
 * Deliberately stricter than the reference in one respect: the Description
 * comes out of a file made by someone else and names the folder that is
 * deleted and recreated, so one that is not a single path component ("..",
 * ".", absolute, containing '/' or '\\') is refused with SNAPPY_E_INVALID_ARG
 * before the file system is touched (the reference passes it to Path.Combine
 * and Directory.Delete as it is, McdfManager.cs:93-107). */
int snappy_mcdf_import_to_dir(const char *mcdf_path, const char *working_dir,
                              char *snapshot_dir_out, uint64_t out_cap);

/* ------------------------------------------------- the write side (archives) */

/* McdfHeader.WriteToStream (Snappy/Features/Mcdf/McdfHeader.cs:14-24) with
 * McdfData.ToByteArray (McdfData.cs:15-18): "MCDF", version 1, int32 length,
 * compact JSON {Description, GlamourerData, CustomizePlusData, ManipulationData,
 * Files:[{GamePaths:[...], Length}]}.  Strings are UTF-8 (NULL = empty).
 * game_paths holds all files' paths one file after the other, paths_per_file[i]
 * of them for file i.  *out_len receives the size; SNAPPY_E_CAPACITY when
 * out_cap is smaller (nothing written).  Host only. */
int snappy_mcdf_write_header(const char *description, const char *glamourer_data,
                             const char *customize_plus_data, const char *manipulation_data,
                             const char *const *game_paths, const uint32_t *paths_per_file,
                             const int64_t *lengths, uint64_t n_files, uint8_t *out,
                             uint64_t out_cap, uint64_t *out_len);

/* The inverse of snappy_lz4net_decode: what writing `in` through
 * `new LZ4Stream(fs, LZ4StreamMode.Compress, ...)` produces in framing (the
 * stream McdfManager.cs:72-80 reads back): one chunk per block_size input bytes
 * (0 = lz4net's default, 1 MiB), each compressed as one LZ4 block on the GPU and
 * stored uncompressed when it does not shrink.  The blocks are valid LZ4 and
 * decode to `in`; they are not byte-identical with lz4net's own compressor.
 * The reference itself never writes an archive (WriteToStream has no caller).
 * snappy_lz4net_encode_bound gives a sufficient out_cap; with a smaller one the
 * call returns SNAPPY_E_CAPACITY and *out_len = the size needed. */
int snappy_lz4net_encode_bound(uint64_t in_len, uint32_t block_size, uint64_t *bound);
int snappy_lz4net_encode(const uint8_t *in, uint64_t in_len, uint32_t block_size,
                         uint8_t *out, uint64_t out_cap, uint64_t *out_len);

/* --------------------------------------------------------------- profiling */

/* Kernel timing of the most recent compute call made by this thread, measured
 * with HIP events recorded on the stream the kernels ran on.  kernel: 0 = SHA-1
 * kernels, 1 = LZ4 decode (or encode) kernel.  Returns milliseconds, or a negative value
 * when that kernel did not run.  snappy_profile_enable(1) turns recording on
 * (off by default).  No counterpart in the reference. */
void snappy_profile_enable(int on);
float snappy_profile_last_ms(int kernel);

/* How the most recent SHA-1 batch of this thread was cut into execution paths
 * (recorded only while profiling is on).  path_end[k] is the end, in the
 * batch's size-sorted order, of path k's range: 64, 16, 4 and 1 lanes of a
 * wavefront per file; path_end[3] is the number of files.  planned_ms is the
 * finishing time the planner's cost model expected.  Returns
 * SNAPPY_E_INVALID_ARG when nothing was recorded.  No counterpart in the
 * reference. */
int snappy_sha1_last_plan(uint32_t path_end[4], double *planned_ms);

/* The cost table the SHA-1 planner uses on `device`: per execution path (64, 16,
 * 4 lanes per file, one lane per file) the nanoseconds a file's chain advances
 * per 64-byte block and the SIMD time a block costs.  *measured (may be NULL) is
 * 1 once the table's level has been set from this device's own chain time (the
 * first hashing call measures it, about 7 ms once per device and process;
 * SNAPPY_SHA1_CALIBRATE=0 keeps the built-in figures).  No counterpart in the
 * reference. */
int snappy_sha1_cost_model(int device, double chain_ns[4], double simd_ns[4], int *measured);

#ifdef __cplusplus
}
#endif
#endif /* SNAPPYHASH_H */

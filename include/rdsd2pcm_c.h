/* rdsd2pcm_c.h -- C ABI of the file-level host driver (include/rdsd2pcm.hpp flattened), exported by the
 * same libdsd2dxd_amd.so as the engine ABI (dsd2dxd_amd.h).
 *
 * This is what a `rdsd2pcm`-named shim crate binds when it wants the containers, sinks and tag handling
 * from this repository as well, so that dsd2dxd's src/main.rs builds unchanged against it
 * (integration/rdsd2pcm-shim/).  One function per crate item that
 * src/main.rs and src/bin/dsd_levels/main.rs use:
 *
 *   d2dh_new                <- Rdsd2Pcm::new             (src/main.rs:325-342,376-392)
 *   d2dh_from_container     <- Rdsd2Pcm::from_container  (src/main.rs:362-373)
 *   d2dh_new_level_check    <- Rdsd2Pcm::new_level_check (src/bin/dsd_levels/main.rs:214-223)
 *   d2dh_do_conversion      <- do_conversion(&CANCEL_FLAG, sender)   (src/main.rs:345,429)
 *   d2dh_check_level        <- check_level(&CANCEL, sender) -> f32   (src/bin/dsd_levels/main.rs:252)
 *   d2dh_file_name          <- file_name()               (src/main.rs:398)
 *   d2dh_find_dsd_files     <- find_dsd_files(&paths, recurse)       (src/main.rs:275)
 *   d2dh_is_container       <- DsdFileFormat::from(&path).is_container()   (src/main.rs:361)
 *
 * Enumerations travel as the CLI's own letters: dither 'T' 'R' 'F' 'X' (src/main.rs:172-175), format
 * 'I' 'P' (:185-186), endianness 'L' 'M' (:194-196), filter 'E' 'X' 'D' 'C' (:200-204), output
 * 'S' 'A' 'C' 'W' 'F' (:208-213).  Every function returns 0 or a negative D2D_ERR_* code; the message
 * (the text the reference would carry in its Err(...)) is in d2dh_last_error() of the calling thread.
 * One object is created, used and freed on one thread, like one Rdsd2Pcm on one Rayon worker. */
#ifndef RDSD2PCM_C_H
#define RDSD2PCM_C_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct d2dh_conv d2dh_conv;
typedef void (*d2dh_progress_fn)(void* user, float percent);     /* the last call passes exactly 100.0f */
typedef void (*d2dh_path_fn)(void* user, const char* path);

int d2dh_new(uint32_t bit_depth, uint32_t output, double level_db, uint32_t output_rate, const char* out_dir /* NULL = next to the input */,
             uint32_t dither, uint32_t fmt, uint32_t endian, uint32_t dsd_rate, uint32_t block_size, uint32_t channels,
             uint32_t filter, int append_rate, const char* base_dir, const char* in_path /* NULL = stdin */, d2dh_conv** out);
int d2dh_from_container(uint32_t bit_depth, uint32_t output, double level_db, uint32_t output_rate, const char* out_dir,
                        uint32_t dither, uint32_t filter, int append_rate, const char* base_dir, const char* path, d2dh_conv** out);
int d2dh_new_level_check(uint32_t output_rate, const char* path /* NULL = stdin */, uint32_t fmt, uint32_t endian, uint32_t channels,
                         uint32_t block_size, uint32_t input_rate, d2dh_conv** out);
void d2dh_free(d2dh_conv* c);

/* `cancel` may be NULL; it is polled between blocks (non-zero = stop, the call returns D2D_ERR_CANCELLED
 * with the message "Conversion cancelled").  `progress` may be NULL. */
int d2dh_do_conversion(d2dh_conv* c, const volatile int* cancel, d2dh_progress_fn progress, void* user);
int d2dh_check_level(d2dh_conv* c, const volatile int* cancel, d2dh_progress_fn progress, void* user, float* peak_dbfs);

const char* d2dh_file_name(const d2dh_conv* c);      /* valid until d2dh_free */
const char* d2dh_output_path(const d2dh_conv* c);    /* "" for stdout; known after d2dh_do_conversion */
const char* d2dh_warnings(const d2dh_conv* c);       /* non-fatal findings of the last run, "" if none */
void d2dh_set_device(d2dh_conv* c, int device);      /* which GPU: e.g. rayon::current_thread_index() % n_gpus */
void d2dh_set_seed(d2dh_conv* c, uint64_t seed);
void d2dh_set_tap_bits(d2dh_conv* c, uint32_t bits);   /* 24 (default) or 32: d2d_params.tap_bits, include/dsd2dxd_amd.h */

int d2dh_find_dsd_files(const char* const* paths, size_t n_paths, int recurse, d2dh_path_fn each, void* user);
int d2dh_is_container(const char* path);              /* 1 for .dsf / .dff, else 0 */
const char* d2dh_last_error(void);                    /* thread-local */

#ifdef __cplusplus
}
#endif
#endif

// ire_ffi.mjs -- the ffi-napi binding BASELINE.json's north_star names: every entry point of include/ire.h, equivalent to
// the N-API shim (ire_napi.cc).  `ffi-napi` / `ref-napi` cannot be installed in the build image (no network), so this file is
// never LOADED there; tests/test_abi.py parses it and checks every declaration (name, return type, argument count and kinds)
// against include/ire.h, so a deployment that has the two modules can bind libire.so without compiling the shim.
//   const lib = bindIre('/opt/ire/image_restoration_platform_amd/lib/libire.so');
//   lib.ire_classify.async(engine, rgb, 1, h, w, 3 * w, flags, scores /* Float64Array(7) */, labels, cb)   // libuv pool, like sharp
import ffi from 'ffi-napi';
import ref from 'ref-napi';

const P = ref.refType(ref.types.void);       // any pointer (buffers, opaque handles, structs)
const PP = ref.refType(P);                   // pointer to a handle (ire_engine**, ire_job**, ire_strips**)
const IP = ref.refType(ref.types.int);       // int*
export const IRE_ABI_VERSION = 3;
export function bindIre(libPath) {
  return ffi.Library(libPath, {
    ire_abi_version: ['int', []],
    ire_init: ['int', [P, PP]],
    ire_shutdown: ['void', [P]],
    ire_last_error: ['string', []],
    ire_load_weights: ['int', [P, P, 'size_t']],
    ire_max_batch_for: ['int', [P, 'int', 'int']],
    ire_classify: ['int', [P, P, 'int', 'int', 'int', 'int', P, P, P]],
    ire_restore: ['int', [P, P, 'int', 'int', 'int', P, P, P, P]],
    ire_fuse: ['int', [P, P, 'int', 'int', 'int', 'double', P, P, P]],
    ire_classify_device: ['int', [P, P, 'int', 'int', 'int', P, P, P, P]],
    ire_restore_device: ['int', [P, P, 'int', 'int', 'int', P, P, P, P]],
    ire_fuse_device: ['int', [P, P, 'int', 'int', 'int', 'double', P, P, P]],
    ire_fuse_batch_device: ['int', [P, P, 'int', 'int', 'int', 'int', P, P, P, P]],
    ire_preprocess_plan: ['int', ['int', 'int', 'int', 'int', IP, IP, IP]],
    ire_preprocess: ['int', [P, P, 'int', 'int', 'int', 'int', P, 'int', 'int']],
    ire_preprocess_device: ['int', [P, P, 'int', 'int', 'int', 'int', P, 'int', 'int', P]],
    ire_png_base64_bytes: ['size_t', ['int', 'int']],
    ire_encode_png_base64_device: ['int', [P, P, 'int', 'int', 'int', P, 'size_t', P]],
    ire_encode_png_base64: ['int', [P, P, 'int', 'int', 'int', P, 'size_t']],
    ire_submit: ['int', [P, P, 'int', 'int', 'int', P, PP]],
    ire_poll: ['int', [P, P, 'int', P, P, P]],
    ire_job_release: ['int', [P, P]],
    ire_affinity_plan: ['int', ['string', 'string', P, 'size_t', IP, IP, IP]],
    ire_engine_affinity: ['int', [P, P, 'size_t', IP]],
    ire_restore_tiled_device: ['int', [P, P, 'int', 'int', 'int', P, P, P, P]],
    ire_strips_stats_bytes: ['size_t', ['int', 'int']],
    ire_strips_open: ['int', [P, 'int', 'int', 'int', 'int', 'int', P, PP]],
    ire_strips_close: ['void', [P]],
    ire_strips_num_ops: ['int', [P]],
    ire_strips_set_input: ['int', [P, P, P, P]],
    ire_strips_run_op: ['int', [P, 'int', P, P]],
    ire_strips_pack_halo: ['int', [P, 'int', P, P, P]],
    ire_strips_unpack_halo: ['int', [P, 'int', P, P, P]],
    ire_strips_get_output: ['int', [P, P, P]],
    ire_get_stats: ['int', [P, P]],
    ire_debug_classifier_sums: ['int', [P, 'int', P]],
    ire_debug_capture: ['int', [P, 'int']],
    ire_debug_activation: ['int', [P, 'string', P, P]],
    ire_profile_enable: ['int', [P, 'int']],
    ire_profile_query: ['int', [P, 'string', P, P, P, P]],
    ire_profile_reset: ['int', [P]],
    ire_profile_report: ['int', [P, P, 'size_t', P]],
  });
}

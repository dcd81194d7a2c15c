// ire_ffi.mjs -- the ffi-napi binding BASELINE.json's north_star names, equivalent to ire_napi.cc.
// Not loadable in the build image (ffi-napi cannot be installed offline); kept so a deployment that has
// `ffi-napi` + `ref-napi` can bind libire.so without compiling the shim.  Signatures = include/ire.h.
import ffi from 'ffi-napi';
import ref from 'ref-napi';

const voidPtr = ref.refType(ref.types.void);
export function bindIre(libPath) {
  return ffi.Library(libPath, {
    ire_abi_version: ['int', []],
    ire_init: ['int', [voidPtr /* const ire_config* */, ref.refType(voidPtr) /* ire_engine** */]],
    ire_shutdown: ['void', [voidPtr]],
    ire_last_error: ['string', []],
    ire_classify: ['int', [voidPtr, voidPtr, 'int', 'int', 'int', 'int', voidPtr, voidPtr, voidPtr]],
    ire_restore: ['int', [voidPtr, voidPtr, 'int', 'int', 'int', voidPtr, voidPtr, voidPtr, voidPtr]],
    ire_fuse: ['int', [voidPtr, voidPtr, 'int', 'int', 'int', 'double', voidPtr, voidPtr, voidPtr]],
    ire_submit: ['int', [voidPtr, voidPtr, 'int', 'int', 'int', ref.refType(voidPtr)]],
    ire_poll: ['int', [voidPtr, voidPtr, 'int', voidPtr, voidPtr, voidPtr]],
    // preprocess step in front of the path (imagePreprocess.js:24-91): size rule + orient / fit-inside resize on the GPU
    ire_preprocess_plan: ['int', ['int', 'int', 'int', 'int', ref.refType('int'), ref.refType('int'), ref.refType('int')]],
    ire_preprocess: ['int', [voidPtr, voidPtr, 'int', 'int', 'int', 'int', voidPtr, 'int', 'int']],
  });
}
// usage: lib.ire_classify.async(engine, rgb, 1, h, w, 3 * w, flags, scores, labels, cb)  (libuv pool, like sharp)

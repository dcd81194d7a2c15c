// Contract test of the two seams (mirrors server-node/tests/restoratorService.test.js:18-79 for the
// engine-backed objects).  Usage: node test_adapters.js <case.json>; prints one JSON line.
// The codec is a raw one ("RAW1" + u16 width + u16 height + u8 isJpeg + RGB) because sharp is not installable here.
'use strict';
const fs = require('fs');
const ad = require('./engine_adapters.js');

const rawCodec = {
  decode: async (buf) => {
    if (buf.length < 9 || buf.toString('ascii', 0, 4) !== 'RAW1') throw new Error('Input buffer contains unsupported image format');
    const w = buf.readUInt16LE(4), h = buf.readUInt16LE(6);
    return { data: buf.slice(9, 9 + w * h * 3), width: w, height: h, format: buf[8] ? 'jpeg' : 'png' };
  },
  encode: async (o) => Buffer.concat([Buffer.from('RAW1'), Buffer.from([o.width & 255, o.width >> 8, o.height & 255, o.height >> 8, 0]), o.data]),
};

// minimal RestoratorService-shaped harness (restorator.js:37-172 contract: never throws, envelope fields)
async function restoreLikeReference(classifier, restorer, imageBuffer) {
  const timings = {};
  const t0 = Date.now();
  try {
    let t = Date.now();
    const degradation = await classifier.analyze(imageBuffer);
    timings.classify_ms = Date.now() - t;
    t = Date.now();
    const r = await restorer.restoreImage({ prompt: 'p', images: [imageBuffer], userContext: { userId: 'u' } });
    timings.restore_ms = Date.now() - t;
    return { success: true, degradationAnalysis: degradation, restoredImage: r.base64Image, metadata: r.metadata, timings };
  } catch (error) {
    return { success: false, error: { message: error.message, code: error.code || 'RESTORATION_FAILED' }, timings, total: Date.now() - t0 };
  }
}

(async () => {
  const spec = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
  const out = { loaded: true };
  let engine;
  try {
    engine = ad.createEngine({ weightsPath: spec.weights, maxBatch: 8 });
    out.engine = true;
  } catch (e) {
    out.engine = false;
    out.initError = e.message;
    console.log(JSON.stringify(out));
    return;
  }
  const classifier = ad.createEngineClassifier({ engine, codec: rawCodec });
  const restorer = ad.createEngineRestorer({ engine, codec: rawCodec });
  const img = fs.readFileSync(spec.image);
  // >= 8 calls in flight (the reference keeps 3 per batch / 5 per worker in flight)
  const many = await Promise.all(Array.from({ length: 8 }, () => classifier.analyze(img)));
  out.scores = many[0];
  out.allEqual = many.every((m) => JSON.stringify(m) === JSON.stringify(many[0]));
  const r = await restoreLikeReference(classifier, restorer, img);
  out.success = r.success;
  out.metadata = r.metadata;
  out.restoredSha = require('crypto').createHash('sha256').update(Buffer.from(r.restoredImage, 'base64').slice(9)).digest('hex');
  const bad = await restoreLikeReference(classifier, restorer, Buffer.from('not an image'));
  out.bad = bad;
  if (spec.worker) {
    // the queue worker (restoration_worker.js) over the same two seams: one job through a RestoratorService-shaped object
    const wk = require('./restoration_worker.js');
    const updates = [];
    const restorator = { restore: async ({ imageBuffer, userPrompt, userContext, options = {} }) => {   // restorator.js:37
      const buf = imageBuffer;
      if (typeof buf.length !== 'number') throw new Error('imageBuffer missing');                       // restorator.js:43
      const e = await restoreLikeReference(classifier, restorer, buf);
      return e.success ? Object.assign(e, { enhancedPrompt: 'p', timings: Object.assign({ prompt_ms: 0, total_ms: 1 }, e.timings) })
                       : { success: false, error: { message: e.error.message, code: e.error.code, type: /invalid|unsupported/i.test(e.error.message) ? 'INVALID_INPUT' : 'UNKNOWN_ERROR' }, timings: e.timings, metadata: { failureStage: 'CLASSIFICATION' } };
    } };
    const processor = wk.createJobProcessor({ restorator, jobStore: { update: async (id, patch) => { updates.push(patch.status); } } });
    const good = await processor({ id: 'w1', attemptsMade: 0, data: { userId: 'u', image: img.toString('base64') } });
    let err = null;
    try { await processor({ id: 'w2', attemptsMade: 0, data: { userId: 'u', image: Buffer.from('junk').toString('base64') } }); } catch (e) { err = { message: e.message, unrecoverable: !!e.unrecoverable, type: e.type }; }
    out.worker = { good, err, updates };
  }
  if (spec.concurrent) {
    // 8 in-flight single-image jobs (the reference keeps 3 per batch / 5 per worker in flight, restorator.js:14, design.md:851)
    // must share engine batches: ire_submit on the JS thread, ire_poll on the pool -- counted by the engine itself
    const hl = ad.createEngineHealth({ engine });
    const img8 = fs.readFileSync(spec.image8);                                         // H, W multiples of 8: no padding, classified inside
    const single = await restorer.restoreImage({ prompt: 'p', images: [Buffer.from(img8)] });
    const before = hl.metrics();
    const bufs = Array.from({ length: spec.concurrent }, () => Buffer.from(img8));    // distinct Buffer objects
    const t0 = Date.now();
    const rs = await Promise.all(bufs.map((b) => restorer.restoreImage({ prompt: 'p', images: [b] })));
    const ms = Date.now() - t0;
    const after = hl.metrics();
    out.concurrent = { batches: after.batches - before.batches, images: after.images - before.images, lastBatch: after.lastBatch, ms: ms,
                       allEqual: rs.every((x) => x.base64Image === rs[0].base64Image), sameAsSingle: rs[0].base64Image === single.base64Image,
                       imagesPerSec: after.imagesPerSec, health: await hl.checkEngine() };
    // decode + classify once per job: analyze() then restoreImage() on the SAME Buffer adds no classifier-only call
    const b2 = Buffer.from(img);
    const s0 = hl.metrics();
    const an = await classifier.analyze(b2);
    const rr = await restorer.restoreImage({ prompt: 'p', images: [b2] });
    out.once = { scoresEqual: JSON.stringify(an) === JSON.stringify(out.scores), pixelsEqual: rr.base64Image === r.restoredImage, cached: engine.seen.has(b2),
                 batches: hl.metrics().batches - s0.batches };
  }
  if (spec.cfg0) {
    // BASELINE cfg 0 through the Node seam: one 256 x 256 job, default prompt, no fusion
    const r0 = await restoreLikeReference(classifier, restorer, fs.readFileSync(spec.cfg0));
    out.cfg0 = { success: r0.success, sha: r0.success ? require('crypto').createHash('sha256').update(Buffer.from(r0.restoredImage, 'base64').slice(9)).digest('hex') : null,
                 scores: r0.degradationAnalysis };
  }
  if (spec.textResults) {
    // resultCodec 'png-device': the engine's batcher hands back the base64 text of a device-encoded PNG; restoreImage passes it on as it is
    const engine2 = ad.createEngine({ weightsPath: spec.weights, maxBatch: 8, resultCodec: 'png-device' });
    const restorer2 = ad.createEngineRestorer({ engine: engine2, codec: rawCodec });
    const tr = await restorer2.restoreImage({ prompt: 'p', images: [fs.readFileSync(spec.image8)] });
    out.textResult = { chars: tr.base64Image.length, sha: require('crypto').createHash('sha256').update(tr.base64Image, 'latin1').digest('hex'),
                       head: Buffer.from(tr.base64Image.slice(0, 16), 'base64').toString('latin1').slice(1, 4) };
  }
  if (spec.fuse) {
    const views = spec.fuse.map((f) => fs.readFileSync(f));
    const fr = await restorer.restoreImage({ prompt: 'p', images: views });
    out.fusedLen = Buffer.from(fr.base64Image, 'base64').length;
  }
  console.log(JSON.stringify(out));
})().catch((e) => { console.log(JSON.stringify({ fatal: e.message })); process.exit(1); });

// PCIe- and codec-inclusive rate of the Node seams (tools/host_path_rate.py): N single-image jobs in flight in a closed loop (the
// reference keeps 3 per batch / 5 per worker), raw codec (sharp is not installable in the build image).  Two loops over the same
// pre-built inputs: (1) the SHIM alone -- addon.restoreAsync: Buffer -> ire_submit -> waiter thread ire_poll -> result Buffer;
// (2) the whole seam -- restorer.restoreImage: decode, restoreAsync, encode, base64 string (the reference's result contract,
// restorator.js:108), whose per-job cost on the ONE JS thread (V8's base64 of a raw 3 MB image) is reported beside it.
// usage: node rate_adapters.js <weights> <size> <inflight> <total>   -> one JSON line
'use strict';
const ad = require('./engine_adapters.js');
const rawCodec = {
  decode: async (buf) => ({ data: buf.slice(9), width: buf.readUInt16LE(4), height: buf.readUInt16LE(6), format: buf[8] ? 'jpeg' : 'png' }),
  encode: async (o) => o.data,
};
(async () => {
  const [weights, size, inflight, total] = [process.argv[2], Number(process.argv[3] || 1024), Number(process.argv[4] || 8), Number(process.argv[5] || 64)];
  const engine = ad.createEngine({ weightsPath: weights, maxBatch: 8 });
  const restorer = ad.createEngineRestorer({ engine, codec: rawCodec });
  const health = ad.createEngineHealth({ engine });
  const px = Buffer.alloc(size * size * 3);
  for (let i = 0; i < px.length; i++) px[i] = (i * 2654435761 >>> 24) & 255;
  const img = Buffer.concat([Buffer.from('RAW1'), Buffer.from([size & 255, size >> 8, size & 255, size >> 8, 1]), px]);
  const inputs = Array.from({ length: total }, () => Buffer.from(img));       // every job its own upload Buffer, built before the clock starts
  const flags = Buffer.from([1]);
  async function closedLoop(one) {
    await Promise.all(Array.from({ length: Math.max(inflight, 16) }, (_, i) => one(i % total)));    // warm-up (two engine batches)
    const b0 = health.metrics().batches;
    const t0 = process.hrtime.bigint();
    let started = 0;
    async function lane() { while (started < total) { const i = started; started += 1; await one(i); } }
    await Promise.all(Array.from({ length: inflight }, lane));
    const dt = Number(process.hrtime.bigint() - t0) / 1e9;
    return { imagesPerSec: total / dt, engineBatches: health.metrics().batches - b0, seconds: dt };
  }
  const shim = await closedLoop((i) => engine.addon.restoreAsync(engine.handle, inputs[i].slice(9), 1, size, size, flags, null, 120000));
  const seam = await closedLoop((i) => restorer.restoreImage({ prompt: 'p', images: [inputs[i]] }));
  const t0 = process.hrtime.bigint();
  let n = 0;
  for (let i = 0; i < 8; i++) n += px.toString('base64').length;
  const b64ms = Number(process.hrtime.bigint() - t0) / 1e6 / 8;
  console.log(JSON.stringify({ size, inflight, total, shim, seam, imagesPerSec: seam.imagesPerSec, engineBatches: seam.engineBatches,
    jsThreadBase64MsPerJob: b64ms, jsThreadBound: 1e3 / b64ms, gauge: health.metrics().imagesPerSec, n: n > 0 }));
})().catch((e) => { console.log(JSON.stringify({ fatal: e.message })); process.exit(1); });

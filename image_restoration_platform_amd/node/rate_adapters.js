// PCIe- and codec-inclusive rate of the Node seams (tools/host_path_rate.py): N concurrent single-image restoreImage calls in
// flight (the reference keeps 3 per batch / 5 per worker), raw codec (sharp is not installable in the build image).
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
  const one = () => restorer.restoreImage({ prompt: 'p', images: [Buffer.from(img)] });
  await Promise.all(Array.from({ length: inflight }, one));          // warm-up
  const b0 = health.metrics().batches;
  const t0 = Date.now();
  let started = 0;
  async function lane() { while (started < total) { started += 1; await one(); } }
  await Promise.all(Array.from({ length: inflight }, lane));
  const dt = (Date.now() - t0) / 1e3;
  const m = health.metrics();
  console.log(JSON.stringify({ size, inflight, total, seconds: dt, imagesPerSec: total / dt, engineBatches: m.batches - b0, gauge: m.imagesPerSec }));
})().catch((e) => { console.log(JSON.stringify({ fatal: e.message })); process.exit(1); });

// engine_adapters.js -- the two duck-typed seams of the reference's RestoratorService, backed by the
// MI355X engine through the N-API shim (ire_napi.node -> libire.so, include/ire.h).
//
//   createEngineClassifier(opts) -> { analyze(imageBuffer) -> Promise<{blur,noise,lowLight,compression,scratch,fade,colorShift}> }
//       stands in for ClassifierService            server-node/src/services/classifier.js:40-99
//   createEngineRestorer(opts)   -> { restoreImage({prompt, images, userContext}) -> Promise<{base64Image, metadata}> }
//       stands in for GeminiClient.restoreImage     server-node/src/clients/geminiClient.js:32-97
//
// Wiring in the reference (the swap point, server-node/src/context/services.js:56-59):
//     const restorer = createEngineRestorer({ engine });
//     const svc = new RestoratorService({ geminiClient: restorer, logger });
//     svc.classifier = createEngineClassifier({ engine });      // public mutable property, restorator.js:24
//
// The engine consumes decoded RGB; `decode(buffer) -> Promise<{data:Buffer(RGB), width, height, format}>` and
// `encode({data,width,height}) -> Promise<Buffer>` are injected.  In the reference deployment they are two
// lines of sharp (see defaultCodec below); tests inject a raw codec because sharp is not installable offline.
// CommonJS + no optional chaining on purpose: it must also load on the Node 12 that ships in the build image.
'use strict';
const path = require('path');

const KEYS = ['blur', 'noise', 'lowLight', 'compression', 'scratch', 'fade', 'colorShift']; // classifier.js:62-70

function loadAddon() {
  return require(path.join(__dirname, 'ire_napi.node'));
}

function createEngine(opts) {
  opts = opts || {};
  const addon = opts.addon || loadAddon();
  const lib = opts.libPath || path.join(__dirname, '..', 'lib', 'libire.so');
  // throws Error('service unavailable: ...') when there is no gfx950 device: there is no CPU fallback
  const handle = addon.init(lib, opts.weightsPath || '', opts.deviceIndex || 0, opts.maxBatch || 8, opts.numStreams || 0);
  return { addon, handle };
}

function defaultCodec() {
  const sharp = require('sharp'); // reference dependency (package.json:37); resolved lazily
  return {
    decode: async (buf) => {
      const img = sharp(buf);
      const meta = await img.metadata();
      const raw = await img.removeAlpha().toColourspace('srgb').raw().toBuffer({ resolveWithObject: true });
      return { data: raw.data, width: raw.info.width, height: raw.info.height, format: meta.format };
    },
    encode: async (o) => sharp(o.data, { raw: { width: o.width, height: o.height, channels: 3 } }).png().toBuffer(),
  };
}

function padToMultipleOf8(img) {
  // RestoreNet has three stride-2 levels: replicate-pad H and W up to multiples of 8 (>= 16), crop afterwards
  const W = Math.max(16, Math.ceil(img.width / 8) * 8), H = Math.max(16, Math.ceil(img.height / 8) * 8);
  if (W === img.width && H === img.height) return { data: img.data, width: W, height: H };
  const out = Buffer.alloc(W * H * 3);
  for (let y = 0; y < H; y++) {
    const sy = Math.min(y, img.height - 1);
    for (let x = 0; x < W; x++) {
      const sx = Math.min(x, img.width - 1);
      img.data.copy(out, (y * W + x) * 3, (sy * img.width + sx) * 3, (sy * img.width + sx) * 3 + 3);
    }
  }
  return { data: out, width: W, height: H };
}

function crop(data, W, w, h) {
  if (W === w) return data.slice(0, w * h * 3);
  const out = Buffer.alloc(w * h * 3);
  for (let y = 0; y < h; y++) data.copy(out, y * w * 3, y * W * 3, y * W * 3 + w * 3);
  return out;
}

function createEngineClassifier(opts) {
  const engine = opts.engine;
  const codec = opts.codec || defaultCodec();
  return {
    async analyze(imageBuffer) {
      const img = await codec.decode(imageBuffer); // rejects on undecodable input -> failureStage CLASSIFICATION
      const flags = Buffer.from([img.format === 'jpeg' ? 1 : 0]); // classifier.js:180 branches on the container
      const r = await engine.addon.classifyAsync(engine.handle, img.data, 1, img.height, img.width, flags);
      const out = {};
      KEYS.forEach((k, i) => { out[k] = r.scores[i]; });
      return out;
    },
  };
}

let jobCounter = 0;
function createEngineRestorer(opts) {
  const engine = opts.engine;
  const codec = opts.codec || defaultCodec();
  return {
    async restoreImage(args) {
      const images = args.images;
      if (!images || images.length < 1 || images.length > 3) throw new Error('invalid images: expected 1..3 encoded images');
      const decoded = [];
      for (const b of images) decoded.push(await codec.decode(b));
      const w = decoded[0].width, h = decoded[0].height;
      if (decoded.some((d) => d.width !== w || d.height !== h)) throw new Error('invalid images: fusion views must have identical dimensions');
      const restored = [];
      let W = w, H = h;
      for (const d of decoded) {
        const p = padToMultipleOf8(d);
        W = p.width; H = p.height;
        const flags = Buffer.from([d.format === 'jpeg' ? 1 : 0]);
        const r = await engine.addon.restoreAsync(engine.handle, p.data, 1, p.height, p.width, flags); // classifies inside
        restored.push(r.pixels);
      }
      let pixels = restored[0];
      if (restored.length > 1) {
        if (H < 64 || W < 64) throw new Error('invalid images: fusion needs at least 64x64 pixels');
        const r = await engine.addon.fuseAsync(engine.handle, Buffer.concat(restored), restored.length, H, W, null, -1.0);
        pixels = r.pixels;
      }
      const png = await codec.encode({ data: crop(pixels, W, w, h), width: w, height: h });
      jobCounter += 1;
      return {
        base64Image: png.toString('base64'),
        metadata: { providerRequestId: 'ire-' + process.pid + '-' + jobCounter, billedTokens: null, estimatedCostUsd: 0 },
      };
    },
  };
}

module.exports = { KEYS, createEngine, createEngineClassifier, createEngineRestorer, padToMultipleOf8 };

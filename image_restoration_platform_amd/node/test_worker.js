// Contract test of restoration_worker.js against a miniature BullMQ Worker (same callbacks, retry and back-off rules:
// processor throw => attemptsMade++, 'failed' event, retry after settings.backoffStrategies[type](attemptsMade) unless the
// attempts are used up or the error is an UnrecoverableError).  Usage: node test_worker.js ; prints one JSON line.
'use strict';
const w = require('./restoration_worker.js');

class UnrecoverableError extends Error {}
class FakeWorker {
  constructor(name, processor, opts) { this.name = name; this.processor = processor; this.opts = opts; this.handlers = {}; this.delays = []; }
  on(ev, fn) { (this.handlers[ev] = this.handlers[ev] || []).push(fn); return this; }
  async emit(ev, a, b) { for (const fn of this.handlers[ev] || []) await fn(a, b); }
  async run(job) {                                   // drives one job to its terminal state, no real sleeping
    for (;;) {
      try { const r = await this.processor(job); job.returnvalue = r; await this.emit('completed', job, r); return r; }
      catch (e) {
        job.attemptsMade += 1;
        await this.emit('failed', job, e);
        await new Promise((res) => setImmediate(res));
        if (e instanceof UnrecoverableError || job.attemptsMade >= job.opts.attempts) return null;
        this.delays.push(this.opts.settings.backoffStrategies[job.opts.backoff.type](job.attemptsMade));
      }
    }
  }
}

function harness(script) {                           // script: array of restore() envelopes, consumed one per attempt
  const log = { updates: [], dlq: [], refunds: [], stored: [], calls: [] };
  let i = 0;
  // destructures ONE object exactly as restorator.js:37 does, and touches imageBuffer.length as restorator.js:43 does (a positional
  // call would throw here, as it would in the reference class)
  const restorator = { restore: async ({ imageBuffer, userPrompt, userContext, options = {} }) => {
    log.calls.push({ bytes: imageBuffer.length, prompt: userPrompt, ctx: userContext, options: options });
    return script[Math.min(i++, script.length - 1)];
  } };
  const deps = {
    Worker: FakeWorker, UnrecoverableError: UnrecoverableError, connection: {}, restorator: restorator, rng: () => 0.5,
    now: () => 'T',
    jobStore: { update: async (id, patch) => { log.updates.push(Object.assign({ id: id }, patch)); } },
    deadLetterQueue: { add: async (name, payload) => { log.dlq.push({ name: name, payload: payload }); } },
    creditsService: { refund: async (userId, jobId, n) => { log.refunds.push([userId, jobId, n]); } },
    storeResult: async (job, buf) => { log.stored.push(buf.length); return { gcsResultPath: 'results/' + job.id + '.png', signedResultUrl: 'https://signed/' + job.id }; },
    loadImage: async (ref) => Buffer.from('from:' + ref),
  };
  const made = w.createRestorationWorker(deps);
  return { made: made, log: log };
}
const ok = { success: true, restoredImage: Buffer.from('pixels').toString('base64'), degradationAnalysis: { blur: 0.8 }, enhancedPrompt: 'P',
             timings: { classify_ms: 1, prompt_ms: 0, restore_ms: 12, total_ms: 13 },
             metadata: { providerRequestId: 'ire-1-1', estimatedCostUsd: 0, billedTokens: null, classificationIssues: [{ type: 'blur', confidence: 0.8 }] } };
const bad = (type, message) => ({ success: false, error: { message: message, code: 'ENGINE_X', type: type }, timings: { classify_ms: 1 }, metadata: { failureStage: 'RESTORATION' } });
const job = (id, data) => ({ id: id, data: data, attemptsMade: 0, opts: { attempts: 5, backoff: { type: 'jittered-exponential' } } });

(async () => {
  const out = {};
  out.defaults = w.queueDefaults({});
  out.envDefaults = w.queueDefaults({ JOBS_QUEUE_NAME: 'q2', JOBS_MAX_ATTEMPTS: '7', JOBS_BACKOFF_BASE_MS: '250', JOBS_BACKOFF_JITTER: '0.1' });
  out.backoffMid = [1, 2, 3, 4].map((a) => w.calculateBackoff(a, w.queueDefaults({}), () => 0.5));
  out.backoffLo = w.calculateBackoff(1, w.queueDefaults({}), () => 0);
  out.backoffHi = w.calculateBackoff(1, w.queueDefaults({}), () => 0.999999);
  out.backoffZero = w.calculateBackoff(0, w.queueDefaults({}), () => 0.5);

  let h = harness([ok]);
  out.first = await h.made.worker.run(job('j1', { userId: 'u1', image: Buffer.from('abc').toString('base64'), user_prompt: 'fix', traceparent: '00-aa-bb-01' }));
  out.firstLog = h.log; out.queueName = h.made.worker.name; out.concurrency = h.made.worker.opts.concurrency;

  h = harness([bad('SERVICE_UNAVAILABLE', 'service unavailable: busy'), bad('TIMEOUT', 'timeout'), ok]);
  out.retry = await h.made.worker.run(job('j2', { userId: 'u2', gcs_ref: 'originals/u2/j2' }));
  out.retryLog = h.log; out.retryDelays = h.made.worker.delays;

  h = harness([bad('SERVICE_UNAVAILABLE', 'service unavailable: down')]);
  out.exhausted = await h.made.worker.run(job('j3', { userId: 'u3', image: Buffer.from('x').toString('base64') }));
  out.exhaustedLog = h.log; out.exhaustedDelays = h.made.worker.delays;

  h = harness([bad('INVALID_INPUT', 'invalid input: not an image')]);
  const j4 = job('j4', { userId: 'u4', image: Buffer.from('x').toString('base64') });
  out.invalid = await h.made.worker.run(j4);
  out.invalidLog = h.log; out.invalidAttempts = j4.attemptsMade;

  h = harness([ok]);
  const j5 = job('j5', { userId: 'u5' });
  out.empty = await h.made.worker.run(j5);
  out.emptyLog = h.log; out.emptyAttempts = j5.attemptsMade;

  let threw = null;
  try { w.createRestorationWorker({ restorator: { restore() {} } }); } catch (e) { threw = e.message; }
  out.needsWorkerClass = threw;
  console.log(JSON.stringify(out));
})().catch((e) => { console.error(e); process.exit(1); });

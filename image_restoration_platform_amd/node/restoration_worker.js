// BullMQ-compatible restoration worker -- SURVEY.md 8(f) row 1: the consumer the reference plans but does not ship
// (docs: .kiro/specs/doppler-backend-infrastructure/design.md:820-884; producer: server-node/src/queues/jobQueue.js).
//
// Nothing here imports bullmq / ioredis / firestore: the deployment passes the real classes in
// (`createRestorationWorker({ Worker, connection, ... })`), the tests pass fakes.  The worker calls
// RestoratorService.restore (restorator.js:37-172) exactly as the design says; with the engine adapters wired in
// (INTEGRATION.md section 1) that call runs on the MI355X.  CommonJS and Node-12 syntax on purpose (see engine_adapters.js).
'use strict';

function num(v, d) { const n = Number(v); return v === undefined || v === null || v === '' || Number.isNaN(n) ? d : n; }

// jobQueue.js:4-9 -- same environment names, same defaults
function queueDefaults(env) {
  const e = env || process.env;
  return {
    queueName: e.JOBS_QUEUE_NAME || 'image-restoration-jobs',
    attempts: num(e.JOBS_MAX_ATTEMPTS, 5),
    backoffBaseMs: num(e.JOBS_BACKOFF_BASE_MS, 1000),
    backoffJitter: num(e.JOBS_BACKOFF_JITTER, 0.3),
    removeOnComplete: num(e.JOBS_REMOVE_ON_COMPLETE, 100),
    removeOnFail: num(e.JOBS_REMOVE_ON_FAIL, 500),
    deadLetterName: e.JOBS_DLQ_NAME || 'image-restoration-dlq',
    concurrency: num(e.JOBS_WORKER_CONCURRENCY, 5),            // design.md:851
  };
}

// The 'jittered-exponential' strategy of jobQueue.js:37-45: base * 2^(attempt-1), uniformly jittered by +-ratio, rounded, >= 0.
// `rng` is injectable so tests are deterministic (the reference uses Math.random).
function calculateBackoff(attemptsMade, opts, rng) {
  const o = opts || queueDefaults();
  const r = rng || Math.random;
  const center = o.backoffBaseMs * Math.pow(2, Math.max(0, attemptsMade - 1));
  const spread = center * o.backoffJitter;
  return Math.round(Math.max(center - spread + r() * 2 * spread, 0));
}

// error types of restorator.js:241-265 that a retry cannot fix
const NON_RETRYABLE = { INVALID_INPUT: true, AUTHENTICATION_FAILED: true };

function toBuffer(image) {
  if (Buffer.isBuffer(image)) return image;
  if (typeof image === 'string') return Buffer.from(image, 'base64');
  if (image && image.type === 'Buffer' && Array.isArray(image.data)) return Buffer.from(image.data);   // JSON round trip through Redis
  return null;
}

// job.data: { userId, gcs_ref | image, user_prompt, traceparent, tracestate } (design.md:195-199, 820-833)
// deps: restorator = the reference's RestoratorService (restore({imageBuffer, userPrompt, userContext, options}), restorator.js:37); loadImage(gcs_ref, job) -> Buffer; storeResult(job, Buffer) -> {gcsResultPath, signedResultUrl};
//       jobStore {update(jobId, patch)} (the jobs/{jobId} document, design.md:642-665); UnrecoverableError (bullmq's, optional)
function createJobProcessor(deps) {
  const restorator = deps.restorator;
  if (!restorator || typeof restorator.restore !== 'function') throw new Error('restoration worker: restorator with restore() is required');
  const jobStore = deps.jobStore || { update: async () => {} };
  const logger = deps.logger || { info() {}, warn() {}, error() {} };
  const now = deps.now || (() => new Date());

  return async function processRestorationJob(job) {
    const data = job.data || {};
    const trace = { traceparent: data.traceparent, tracestate: data.tracestate };
    const attempt = (job.attemptsMade || 0) + 1;
    await jobStore.update(job.id, { status: 'running', attempt: attempt, updatedAt: now() });

    let buffer = toBuffer(data.image);
    if (!buffer && data.gcs_ref) {
      if (typeof deps.loadImage !== 'function') throw fail('service unavailable: no loader for gcs_ref', 'SERVICE_UNAVAILABLE', 'SERVICE_UNAVAILABLE');
      buffer = await deps.loadImage(data.gcs_ref, job);
    }
    if (!buffer) throw unrecoverable(deps, fail('invalid job payload: neither image nor gcs_ref', 'INVALID_INPUT', 'INVALID_INPUT'));

    // restore() never rejects (restorator.js:37-172): failures come back in the envelope
    // ONE object argument, exactly the reference's signature (restorator.js:37): restore({ imageBuffer, userPrompt, userContext, options })
    const result = await restorator.restore({
      imageBuffer: buffer,
      userPrompt: data.user_prompt,
      userContext: { userId: data.userId, jobId: job.id, traceparent: trace.traceparent, tracestate: trace.tracestate },
      options: {},
    });
    if (!result || result.success !== true) {
      const e = (result && result.error) || {};
      const err = fail(e.message || 'restoration failed', e.code || 'RESTORATION_FAILED', e.type || 'UNKNOWN_ERROR');
      err.failureStage = result && result.metadata && result.metadata.failureStage;
      err.timings = result && result.timings;
      logger.warn('[worker] restoration failed', { jobId: job.id, attempt: attempt, type: err.type, stage: err.failureStage });
      throw NON_RETRYABLE[err.type] ? unrecoverable(deps, err) : err;       // a throw is what makes BullMQ retry (design.md:842-845)
    }

    let stored = {};
    if (typeof deps.storeResult === 'function') stored = (await deps.storeResult(job, Buffer.from(result.restoredImage, 'base64'))) || {};
    const md = result.metadata || {};
    const record = {
      status: 'succeeded',
      timings: result.timings,                                   // {classify_ms, prompt_ms, restore_ms, total_ms}
      degradation: result.degradationAnalysis,
      prompt: result.enhancedPrompt,
      providerRequestId: md.providerRequestId,
      costUsd: md.estimatedCostUsd === undefined ? null : md.estimatedCostUsd,
      gcsResultPath: stored.gcsResultPath,
      signedResultUrl: stored.signedResultUrl,
      updatedAt: now(),
    };
    await jobStore.update(job.id, record);
    logger.info('[worker] job succeeded', { jobId: job.id, attempt: attempt, total_ms: result.timings && result.timings.total_ms });
    // the return value is what BullMQ stores on the job; keep it small (no pixels)
    return { status: 'succeeded', timings: record.timings, providerRequestId: record.providerRequestId, gcsResultPath: record.gcsResultPath,
             classificationIssues: md.classificationIssues };
  };
}

function fail(message, code, type) { const e = new Error(message); e.code = code; e.type = type; return e; }
function unrecoverable(deps, err) {
  err.unrecoverable = true;
  if (typeof deps.UnrecoverableError === 'function') {       // bullmq >= 2: a throw of this class skips the remaining attempts
    const u = new deps.UnrecoverableError(err.message);
    u.code = err.code; u.type = err.type; u.failureStage = err.failureStage; u.unrecoverable = true;
    return u;
  }
  return err;
}

// Terminal failure handling of design.md:857-884: DLQ entry, credit refund, jobs/{id} -> failed.  Idempotent per job id.
function createFailedHandler(deps) {
  const jobStore = deps.jobStore || { update: async () => {} };
  const now = deps.now || (() => new Date());
  const done = new Set();
  return async function onFailed(job, error) {
    if (!job) return false;
    const maxAttempts = (job.opts && job.opts.attempts) || queueDefaults().attempts;
    const terminal = (error && error.unrecoverable) || (job.attemptsMade || 0) >= maxAttempts;
    if (!terminal) {                                            // back to 'queued' (state machine, design.md:912-933)
      await jobStore.update(job.id, { status: 'queued', attempt: job.attemptsMade, updatedAt: now() });
      return false;
    }
    if (done.has(job.id)) return true;
    done.add(job.id);
    if (deps.deadLetterQueue) {
      await deps.deadLetterQueue.add('failed-restoration', {
        originalJobId: job.id, data: job.data, attempts: job.attemptsMade,
        error: { message: error && error.message, stack: error && error.stack },
      });
    }
    if (deps.creditsService && job.data) await deps.creditsService.refund(job.data.userId, job.id, 1);
    await jobStore.update(job.id, { status: 'failed', updatedAt: now(),
                                    error: { code: (error && error.code) || 'UNKNOWN_ERROR', message: error && error.message } });
    return true;
  };
}

// Worker settings that make a BullMQ Worker honour the producer's backoff type (jobQueue.js:58-62).  bullmq <= 3 reads
// settings.backoffStrategies[type]; bullmq >= 4 reads settings.backoffStrategy(attemptsMade, type, ...): provide both.
function workerOptions(deps) {
  const d = Object.assign(queueDefaults(deps.env), deps.queue || {});
  const strategy = (attemptsMade) => calculateBackoff(attemptsMade, d, deps.rng);
  return {
    connection: deps.connection,
    concurrency: deps.concurrency || d.concurrency,
    settings: { backoffStrategies: { 'jittered-exponential': strategy }, backoffStrategy: (attemptsMade) => strategy(attemptsMade) },
  };
}

function createRestorationWorker(deps) {
  if (typeof deps.Worker !== 'function') throw new Error('restoration worker: pass bullmq\'s Worker class as deps.Worker');
  const d = Object.assign(queueDefaults(deps.env), deps.queue || {});
  const processor = deps.processor || createJobProcessor(deps);
  const worker = new deps.Worker(d.queueName, processor, workerOptions(deps));
  const onFailed = createFailedHandler(deps);
  worker.on('failed', (job, error) => { onFailed(job, error).catch((e) => (deps.logger || console).error('[worker] failed-handler error', { error: e && e.message })); });
  worker.on('error', (error) => (deps.logger || console).error('[worker] error', { error: error && error.message }));
  return { worker: worker, processor: processor, onFailed: onFailed, options: d };
}

module.exports = { queueDefaults, calculateBackoff, createJobProcessor, createFailedHandler, createRestorationWorker, workerOptions, NON_RETRYABLE };

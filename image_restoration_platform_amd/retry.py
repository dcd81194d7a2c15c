"""Host-side mirror of the reference's retry / queue back-off policy.

  exponential_backoff  <- server-node/src/utils/retry.js:1-47   (provider call: 3 attempts, 500 ms * 2^(n-1) +-30 %)
  calculate_backoff    <- server-node/src/queues/jobQueue.js:37-45 (BullMQ 'jittered-exponential': round(BASE*2^(n-1) +-30 %))
  QUEUE_DEFAULTS       <- jobQueue.js:4-9,56-66 (queue name, attempts 5, retention)
Both take an injectable rng so tests are deterministic (the reference uses Math.random).
"""
import os
import random
import time

QUEUE_DEFAULTS = {
    "name": os.environ.get("JOBS_QUEUE_NAME", "image-restoration-jobs"),
    "attempts": int(float(os.environ.get("JOBS_MAX_ATTEMPTS", "5"))),
    "base_delay_ms": float(os.environ.get("JOBS_BACKOFF_BASE_MS", "1000")),
    "jitter": float(os.environ.get("JOBS_BACKOFF_JITTER", "0.3")),
    "remove_on_complete": int(float(os.environ.get("JOBS_REMOVE_ON_COMPLETE", "100"))),
    "remove_on_fail": int(float(os.environ.get("JOBS_REMOVE_ON_FAIL", "500"))),
}


def calculate_delay(base_delay, attempt, factor, jitter, rng=random.random):
    """retry.js:1-10"""
    delay = base_delay * (factor ** (attempt - 1))
    if not jitter:
        return delay
    j = delay * jitter
    lo, hi = delay - j, delay + j
    return max(0, lo + rng() * (hi - lo))


def exponential_backoff(fn, attempts=3, min_delay_ms=500, factor=2, jitter=0.3, on_retry=None, rng=random.random,
                        sleep=time.sleep):
    """retry.js:12-47 -- returns fn()'s value or re-raises the last error after `attempts` tries."""
    if not callable(fn):
        raise TypeError("fn must be a function")
    last = None
    for attempt in range(1, attempts + 1):
        try:
            return fn()
        except Exception as e:  # noqa: BLE001
            last = e
            if attempt == attempts:
                break
            delay = calculate_delay(min_delay_ms, attempt, factor, jitter, rng)
            if on_retry:
                on_retry(e, {"attempt": attempt, "nextDelayMs": delay})
            sleep(delay / 1000.0)
    raise last


def calculate_backoff(attempts_made, base_delay_ms=None, jitter_ratio=None, rng=random.random):
    """jobQueue.js:37-45"""
    base = QUEUE_DEFAULTS["base_delay_ms"] if base_delay_ms is None else base_delay_ms
    jr = QUEUE_DEFAULTS["jitter"] if jitter_ratio is None else jitter_ratio
    exponent = max(0, attempts_made - 1)
    base_delay = base * (2 ** exponent)
    j = base_delay * jr
    lo, hi = base_delay - j, base_delay + j
    delay = rng() * (hi - lo) + lo
    return int(_js_round(max(delay, 0)))


def _js_round(x):
    import math
    return math.floor(x + 0.5)  # Math.round: half up

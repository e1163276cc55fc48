"""Ordered raw-PCM streaming of synthesised sentences.

The reference's `--raw-stream` mode (`larynx/__main__.py:229-268,335-337`) keeps
synthesis off the output path: worker threads synthesise sentences, the main
loop puts each sentence's int16 bytes on a bounded queue (default 5,
`__main__.py:547-551`) and a writer thread drains it to stdout in sentence
order.  Here the workers are host threads that each keep one batch-1 call in
flight on the GPU (the engine gives every in-flight call its own HIP streams),
so the first sentence's audio is written while later ones are still computing.
"""
from __future__ import annotations

import collections
import queue
import threading
import time
import typing
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import sentence_task

Sentence = typing.Union[typing.Tuple[str, typing.Sequence[int]], typing.Tuple[str, typing.Sequence[int], int, int]]


class RawStreamStats(typing.NamedTuple):
    sentences: int
    samples: int
    seconds_to_first_audio: float
    seconds_total: float


def stream_raw_pcm(sentences: typing.Iterable[Sentence], tts_model, vocoder_model, sink: typing.BinaryIO,
                   tts_settings=None, vocoder_settings=None, max_thread_workers: int = 2,
                   raw_stream_queue_size: int = 5, max_pending: typing.Optional[int] = None) -> RawStreamStats:
    """Synthesise `sentences` — `(text, phoneme_ids)` or `(text, phoneme_ids,
    pause_before_ms, pause_after_ms)` — and write 16-bit mono PCM to `sink` in
    sentence order.  `max_thread_workers=2` is the reference's raw-stream default
    ("faster time to first audio", `__main__.py:236-238`).  `max_pending` bounds
    how many sentences are submitted ahead of the writer (default: workers +
    queue size), so an unbounded sentence source cannot pile audio up in memory.
    A failed sentence re-raises here after the writer has been shut down."""
    audio_settings = getattr(tts_model, "audio_settings", None)
    max_pending = max_pending or (max_thread_workers + raw_stream_queue_size)
    raw: "queue.Queue[typing.Optional[bytes]]" = queue.Queue(maxsize=raw_stream_queue_size)
    sink_error: typing.List[BaseException] = []

    def writer():
        while True:
            chunk = raw.get()
            if chunk is None:
                return
            if sink_error:
                continue  # keep draining so producers never block on a dead sink
            try:
                sink.write(chunk)
                sink.flush()
            except BaseException as e:  # noqa: BLE001 - reported to the caller below
                sink_error.append(e)

    t0 = time.perf_counter()
    first = None
    n_sent = n_samples = 0
    wt = threading.Thread(target=writer, daemon=True)
    wt.start()
    pending: typing.Deque = collections.deque()
    for m in (tts_model, vocoder_model):  # a worker per pool thread (+ a spare) up front: calls spread evenly over the hardware queues
        eng = getattr(m, "engine", None)
        if eng is not None and hasattr(eng, "ensure_workers") and max_thread_workers and max_thread_workers >= 2:
            eng.ensure_workers(int(max_thread_workers) + 1)
    try:
        with ThreadPoolExecutor(max_workers=max_thread_workers) as pool:

            def drain_one():
                nonlocal first, n_sent, n_samples
                audio = pending.popleft().result()
                if first is None:
                    first = time.perf_counter() - t0
                n_sent += 1
                n_samples += int(audio.shape[-1])
                raw.put(np.ascontiguousarray(audio, np.int16).tobytes())
                if sink_error:
                    raise sink_error[0]

            for item in sentences:
                text, ids = item[0], np.asarray(item[1], np.int64)
                before, after = (int(item[2]), int(item[3])) if len(item) >= 4 else (0, 0)
                pending.append(pool.submit(sentence_task, text, ids, audio_settings, tts_model, tts_settings,
                                           vocoder_model, vocoder_settings, before, after))
                while len(pending) >= max_pending:
                    drain_one()
            while pending:
                drain_one()
    finally:
        for f in pending:
            f.cancel()
        raw.put(None)
        wt.join()
    if sink_error:
        raise sink_error[0]
    return RawStreamStats(n_sent, n_samples, first if first is not None else 0.0, time.perf_counter() - t0)

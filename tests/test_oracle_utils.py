"""Known-answer tests of the reference's state containers restated against the NumPy oracle
(/root/reference/tests/utils_test.py:70-301, 585-606).  tests/test_gpu_store.py replays the same sequences
through the device-resident implementation."""
import collections

import numpy as np
import pytest

from oracle import utils_np

I32 = utils_np.Spec((), np.int32)

FULL_SEQ = [  # tests/utils_test.py:84-111: (should_reset, env_id, value), batched by 3
    (False, 0, 10), (False, 2, 30), (False, 1, 20), (False, 0, 11), (False, 2, 31), (False, 3, 40),
    (False, 0, 12), (False, 2, 32), (False, 3, 41), (False, 0, 13), (False, 1, 21), (True, 2, 33),
    (False, 0, 14), (False, 2, 34), (False, 3, 42), (False, 0, 15), (False, 1, 22), (False, 2, 35),
    (False, 0, 16), (False, 1, 23), (False, 2, 36)]
FULL_EXPECT = [([], []), ([], []), ([], []), ([0], [[10, 11, 12, 13]]), ([], []), ([], []),
               ([0, 1, 2], [[13, 14, 15, 16], [20, 21, 22, 23], [33, 34, 35, 36]])]

OVERLAP_SEQ = [  # tests/utils_test.py:199-222, batched by 2
    (False, 0, 10), (False, 1, 20), (False, 0, 11), (False, 1, 21), (False, 0, 12), (True, 1, 22),
    (False, 0, 13), (False, 1, 23), (False, 0, 14), (False, 1, 24), (True, 0, 15), (False, 1, 25),
    (False, 0, 16), (False, 1, 26), (False, 0, 17), (False, 1, 27)]
OVERLAP_EXPECT = [([], []), ([], []), ([0], [[0, 0, 10, 11, 12]]), ([], []),
                  ([0, 1], [[10, 11, 12, 13, 14], [0, 0, 22, 23, 24]]), ([], []), ([1], [[22, 23, 24, 25, 26]]),
                  ([0], [[0, 0, 15, 16, 17]])]


def replay(store, seq, batch, reset_fn=None, append_fn=None):
  out = []
  for i in range(0, len(seq) - len(seq) % batch, batch):
    chunk = seq[i:i + batch]
    ids = np.array([c[1] for c in chunk], np.int32)
    vals = np.array([c[2] for c in chunk], np.int32)
    rs = ids[np.array([c[0] for c in chunk])]
    store.reset(rs)
    out.append(store.append(ids, vals))
  return out


def test_unroll_store_full():
  got = replay(utils_np.UnrollStore(4, 3, I32), FULL_SEQ, 3)
  for (ids, un), (eids, eun) in zip(got, FULL_EXPECT):
    np.testing.assert_array_equal(ids, np.array(eids, np.int64))
    np.testing.assert_array_equal(un.reshape(-1, 4), np.array(eun, np.int32).reshape(-1, 4))


def test_unroll_store_overlap_2():
  got = replay(utils_np.UnrollStore(2, 2, I32, num_overlapping_steps=2), OVERLAP_SEQ, 2)
  for (ids, un), (eids, eun) in zip(got, OVERLAP_EXPECT):
    np.testing.assert_array_equal(ids, np.array(eids, np.int64))
    np.testing.assert_array_equal(un.reshape(-1, 5), np.array(eun, np.int32).reshape(-1, 5))


def test_unroll_store_duplicate_ids():
  store = utils_np.UnrollStore(2, 3, I32)
  with pytest.raises(ValueError):
    store.append(np.array([1, 1]), np.array([42, 43], np.int32))      # utils_test.py:72-79 (ids [2,2] there)


def test_unroll_store_structure():
  nt = collections.namedtuple('named_tuple', 'x y')                    # utils_test.py:162-189
  store = utils_np.UnrollStore(2, 10, nt(I32, I32))
  for _ in range(10):
    ids, un = store.append(np.arange(2), nt(np.zeros(2, np.int32), np.zeros(2, np.int32)))
    assert len(ids) == 0 and un.x.shape == (0, 11)
  ids, un = store.append(np.arange(2), nt(np.zeros(2, np.int32), np.zeros(2, np.int32)))
  np.testing.assert_array_equal(ids, [0, 1])
  assert un.x.shape == (2, 11) and un.y.shape == (2, 11)


def test_aggregator():
  agg = utils_np.Aggregator(4, I32)                                    # utils_test.py:276-286
  np.testing.assert_array_equal(agg.read([0, 1, 2, 3]), [0, 0, 0, 0])
  agg.add([0, 1], np.array([42, 43], np.int32))
  np.testing.assert_array_equal(agg.read([0, 1, 2, 3]), [42, 43, 0, 0])
  agg.reset([0])
  np.testing.assert_array_equal(agg.read([0, 1, 2, 3]), [0, 43, 0, 0])
  agg.replace([0, 2], np.array([1, 2], np.int32))
  np.testing.assert_array_equal(agg.read([0, 1, 2, 3]), [1, 43, 2, 0])


def test_batch_apply_and_time_major():
  a = np.array([[[0, 1], [2, 3]], [[4, 5], [6, 7]]])                  # utils_test.py:291-301
  b = np.array([[[8, 9], [10, 11]], [[12, 13], [14, 15]]])
  s, m = utils_np.batch_apply(lambda x, y: (x.sum(-1), y.max(-1)), (a, b))
  np.testing.assert_array_equal(s, [[1, 5], [9, 13]])
  np.testing.assert_array_equal(m, [[9, 11], [13, 15]])
  x = {'a': np.array([[1, 2], [3, 4]]), 'b': np.array([[1], [2]])}    # utils_test.py:587-606
  tm = utils_np.make_time_major(x)
  np.testing.assert_array_equal(tm['a'], [[1, 3], [2, 4]])
  np.testing.assert_array_equal(tm['b'], [[1, 2]])


# ---- PrioritizedReplay (tests/utils_test.py:304-405) ---- #
def _replay_known_answers(make, to_np, uniforms):
  """The reference's PrioritizedReplayTest cases, parameterised over the implementation under test."""
  import collections as c
  rb = make(2, utils_np.Spec((), np.int32), .5)                                  # test_simple :306-323
  np.testing.assert_array_equal(to_np(rb.insert(np.array([1, 2], np.int32), np.array([1., 1.], np.float32))), [0, 1])
  idx, w, vals = rb.sample(2, .5, uniforms(2))
  np.testing.assert_array_equal(to_np(idx) + 1, to_np(vals)); np.testing.assert_array_equal(to_np(w), [1., 1.])
  idx, w, vals = rb.sample(2, 0)
  np.testing.assert_array_equal(to_np(idx) + 1, to_np(vals)); np.testing.assert_array_equal(to_np(w), [1., 1.])
  rb = make(2, utils_np.Spec((), np.int32), .5)                                  # test_update_priorities :339-354
  rb.insert(np.array([1, 2], np.int32), np.array([1., 1.], np.float32))
  rb.update_priorities(np.array([0]), np.array([100.], np.float32))
  idx, w, vals = rb.sample(2, .5, np.array([0.1, 0.7], np.float32))
  np.testing.assert_array_equal(to_np(idx), [0, 0]); np.testing.assert_array_equal(to_np(vals), [1, 1])
  np.testing.assert_array_equal(to_np(w), [1., 1.])
  rb = make(2, utils_np.Spec((), np.int32), .5)                                  # test_initial_priorities :356-369
  rb.insert(np.array([1, 2], np.int32), np.array([0.1, 0.9], np.float32))
  _, _, vals = rb.sample(1000, 1, uniforms(1000))
  cnt = c.Counter(to_np(vals).tolist())
  assert 1000 * 0.1 * 0.7 < cnt[1] < 1000 * 0.1 * 1.3
  for is_exp, p_exp, expected in (                                               # test_importance_sampling_weights1/2
      (1, 1, np.array([(0.3 + 0.9) / 0.3, (0.3 + 0.9) / 0.9])),
      (.3, .7, np.array([(0.3 ** .7 + 0.9 ** .7) / 0.3 ** .7, (0.3 ** .7 + 0.9 ** .7) / 0.9 ** .7]) ** .3)):
    rb = make(2, utils_np.Spec((), np.int32), is_exp)
    rb.insert(np.array([0, 1], np.int32), np.array([0.3, 0.9], np.float32))
    _, w, vals = rb.sample(100, p_exp, uniforms(100))
    expected = expected / expected.max()
    w, vals = to_np(w), to_np(vals)
    assert set(vals.tolist()) == {0, 1}
    for v in (0, 1):
      np.testing.assert_allclose(w[vals == v], expected[v], rtol=1e-5)
  # wrap-around FIFO + nests (:325-337)
  nt = collections.namedtuple('nt', 'a b')
  rb = make(3, nt(utils_np.Spec((), np.int32), utils_np.Spec((2,), np.int64)), .5)
  for k in range(5):
    ids = rb.insert(nt(np.array([k], np.int32), np.array([[k, -k]], np.int64)), np.array([1.], np.float32))
    assert to_np(ids).tolist() == [k % 3]
  _, _, vals = rb.sample(64, 1, uniforms(64))
  a, b = to_np(vals.a), to_np(vals.b)
  assert set(a.tolist()) <= {2, 3, 4} and np.array_equal(b[:, 0], a) and np.array_equal(b[:, 1], -a)


def test_prioritized_replay_oracle():
  rng = np.random.default_rng(5)
  _replay_known_answers(utils_np.PrioritizedReplay, np.asarray, lambda n: rng.uniform(size=n).astype(np.float32))

"""tf.train.Checkpoint files (tensor bundle = LevelDB-format index + raw data shard): reader / writer round trip, the
table format's fixed points (magic, CRC32C known answers, prefix compression across restart points, multi-block index),
and the reference agents' object-graph key names (dmlab/networks.py:26-89, atari/networks.py:232-252)."""
import os
import struct

import numpy as np
import pytest

from seed_rl_amd import tf_checkpoint as tc


def test_crc32c_known_answers():
  # RFC 3720 B.4 test vectors
  assert tc.crc32c(b'\x00' * 32) == 0x8A9136AA
  assert tc.crc32c(b'\xff' * 32) == 0x62A8AB43
  assert tc.crc32c(bytes(range(32))) == 0x46DD794E
  assert tc.crc32c(b'123456789') == 0xE3069283
  big = bytes(np.random.default_rng(0).integers(0, 256, 100000).astype(np.uint8))
  assert tc.crc32c(big) == tc.crc32c(big[50000:], tc.crc32c(big[:50000]))          # library routine, continued


def test_bundle_round_trip(tmp_path):
  rng = np.random.default_rng(1)
  tensors = {}
  for i in range(150):                                   # > 2 data blocks, shared key prefixes, restart points
    tensors['agent/_stacks/%d/_conv/kernel/.ATTRIBUTES/VARIABLE_VALUE' % i] = rng.normal(size=(3, 3, 2, 4)).astype(np.float32)
  tensors['optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE'] = np.asarray(12345, np.int64)
  tensors['a'] = np.zeros((0, 3), np.float32)
  tensors['big'] = rng.normal(size=(300, 1000)).astype(np.float32)
  prefix = str(tmp_path / 'ckpt-7')
  tc.write_checkpoint(prefix, tensors)
  raw = open(prefix + '.index', 'rb').read()
  assert struct.unpack('<Q', raw[-8:])[0] == 0xdb4775248b80fb57 and len(raw) > 48
  back = tc.read_checkpoint(prefix)
  assert sorted(back) == sorted(tensors)
  for k, v in tensors.items():
    assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v)
  assert list(tc.read_index(prefix + '.index'))[0] == b''        # the header entry sorts first
  # a flipped payload byte is caught by the entry checksum
  data = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
  data[100] ^= 1
  open(prefix + '.data-00000-of-00001', 'wb').write(bytes(data))
  with pytest.raises(ValueError, match='checksum'):
    tc.read_checkpoint(prefix)


def test_snappy_block_decoder():
  # literal "abcd", copy (offset 4, length 4), literal "xy": 'abcdabcdxy'
  comp = bytes([10, (4 - 1) << 2, ord('a'), ord('b'), ord('c'), ord('d'), (0 << 5) | ((4 - 4) << 2) | 1, 4,
                (2 - 1) << 2, ord('x'), ord('y')])
  assert tc._snappy_uncompress(comp) == b'abcdabcdxy'


class _FakeAgent(object):
  pass


def test_reference_key_names():
  from seed_rl_amd import networks
  a = _FakeAgent.__new__(networks.ImpalaDeep)            # only the class layout matters here (no device needed)
  a._channels = (16, 32, 32)
  p = tc.reference_variable_paths(a)
  assert len(p) == 39                                    # tests/agents_test.py:45
  assert p['stack1/res_0/conv2d_1/kernel'] == '_stacks/1/_res_convs1/0/kernel'
  assert p['core/recurrent_kernel'] == '_core/recurrent_kernel' and p['baseline/bias'] == '_baseline/bias'
  d = _FakeAgent.__new__(networks.DuelingLSTMDQNNet)
  q = tc.reference_variable_paths(d)
  assert q['body/fc/kernel'] == '_body/layer_with_weights-3/kernel' and q['advantage/head/kernel'].endswith('-1/kernel')
  assert 'advantage/head/bias' not in q


def test_object_graph_reaches_every_key(tmp_path):
  """The TrackableObjectGraph emitted under _CHECKPOINTABLE_OBJECT_GRAPH (agents/vtrace/learner.py:286-296: what
  tf.train.Checkpoint(agent=..., optimizer=...).restore() walks): every VARIABLE_VALUE key of the file is reachable from
  the root by exactly the attribute path its name spells, Adam's slots hang off the optimizer node with the variable
  they belong to, and the scalar string tensor survives the bundle's checksums."""
  keys = {}
  names = ['agent/_stacks/0/_conv/kernel', 'agent/_stacks/0/_conv/bias', 'agent/_core/recurrent_kernel',
           'agent/entropy_cost_param']
  for i, n in enumerate(names):
    keys[n + tc._SUFFIX] = np.full((2, i + 1), i, np.float32) if i < 3 else np.asarray(0.5, np.float32)
    for slot in ('m', 'v'):
      keys['%s/.OPTIMIZER_SLOT/optimizer/%s%s' % (n, slot, tc._SUFFIX)] = keys[n + tc._SUFFIX] * 0
  keys['optimizer/iter' + tc._SUFFIX] = np.asarray(7, np.int64)
  keys['save_counter' + tc._SUFFIX] = np.asarray(1, np.int64)
  graph = tc.object_graph(keys)
  # structure: node 0 has exactly the three top-level children; the optimizer node references 8 slot variables
  assert sorted(c.local_name for c in graph.nodes[0].children) == ['agent', 'optimizer', 'save_counter']
  opt = [n for n in graph.nodes if len(n.slot_variables)]
  assert len(opt) == 1 and len(opt[0].slot_variables) == 2 * len(names)
  assert sorted(set(s.slot_name for s in opt[0].slot_variables)) == ['m', 'v']
  keys[tc.OBJECT_GRAPH_KEY] = graph.SerializeToString()
  prefix = str(tmp_path / 'ckpt-1')
  tc.write_checkpoint(prefix, keys)
  back = tc.read_checkpoint(prefix)
  assert back[tc.OBJECT_GRAPH_KEY] == keys[tc.OBJECT_GRAPH_KEY]
  walked = tc.walk_object_graph(back[tc.OBJECT_GRAPH_KEY])
  want = {k: k[:-len(tc._SUFFIX)].split('/') for k in keys if k.endswith(tc._SUFFIX)}
  assert walked == want
  # a flipped byte in the string tensor is caught by its checksum
  data = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
  data[5] ^= 0x40
  open(prefix + '.data-00000-of-00001', 'wb').write(bytes(data))
  with pytest.raises(ValueError):
    tc.read_checkpoint(prefix)

"""Reader / writer of TensorFlow checkpoint files (the "tensor bundle" that tf.train.Checkpoint / CheckpointManager
write) and the variable-name map of the reference agents -- SURVEY.md 8(f) rank 4: a reference checkpoint
(/root/reference/agents/vtrace/learner.py:286-296: tf.train.Checkpoint(agent=agent, optimizer=optimizer);
agents/r2d2/learner.py:646-647: + target_agent) can be loaded into the agents here.  The WRITER emits the same keys,
dtypes and shapes for the VARIABLE_VALUE tensors and (r4) a serialized TrackableObjectGraph under
`_CHECKPOINTABLE_OBJECT_GRAPH`, derived from those keys: one node per attribute path, `VARIABLE_VALUE` attributes with
their checkpoint keys, Adam's slots as `slot_variables` of the optimizer node, `save_counter` -- the structure
tf.train.Checkpoint.restore() walks.  Restated from TF's published sources like the rest of this module and NOT checked
against a TF-written file or a TF reader ("format unpinned"): files written here round-trip through this module
(tests/test_tf_checkpoint.py walks the emitted graph from the root to every key); whether TensorFlow's restore()
accepts them is untested.

Format (TF 2.4.1; no TensorFlow in this image, so restated from its published sources and NOT checked against a
TF-written file -- "format unpinned"; the writer and the reader here round-trip, tests/test_tf_checkpoint.py):
  <prefix>.index                 an immutable sorted string table (tensorflow/core/lib/io/table*, the LevelDB table
                                 format): data blocks of prefix-compressed (key, value) entries with restart points, an
                                 index block of (last key of block -> BlockHandle), a metaindex block, and a 48-byte footer
                                 ending in the magic 0xdb4775248b80fb57; every block is followed by a 1-byte compression
                                 type (0 none, 1 snappy) and a masked CRC32C.  Key "" -> BundleHeaderProto, every other
                                 key (a tensor name) -> BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}.
  <prefix>.data-00000-of-00001   the raw little-endian tensor bytes at (offset, size).
Object-graph keys of tf.train.Checkpoint: <attribute path>/.ATTRIBUTES/VARIABLE_VALUE, e.g.
  agent/_stacks/0/_conv/kernel/.ATTRIBUTES/VARIABLE_VALUE, optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE and, for Adam's
  slots, <variable path>/.OPTIMIZER_SLOT/optimizer/m/.ATTRIBUTES/VARIABLE_VALUE (lists by index, Sequential children
  as layer_with_weights-<i>).
"""
import collections
import os
import re
import struct

import numpy as np

from seed_rl_amd import tf_wire

BundleHeaderProto = tf_wire.message_class('tensorflow.BundleHeaderProto')
BundleEntryProto = tf_wire.message_class('tensorflow.BundleEntryProto')
TrackableObjectGraph = tf_wire.message_class('tensorflow.TrackableObjectGraph')
OBJECT_GRAPH_KEY = '_CHECKPOINTABLE_OBJECT_GRAPH'
_MAGIC = 0xdb4775248b80fb57
_SUFFIX = '/.ATTRIBUTES/VARIABLE_VALUE'
DT_FLOAT, DT_INT32, DT_INT64, DT_STRING, DT_BOOL, DT_DOUBLE, DT_UINT8 = 1, 3, 9, 7, 10, 2, 4
_NP = {DT_FLOAT: np.float32, DT_DOUBLE: np.float64, DT_INT32: np.int32, DT_INT64: np.int64, DT_BOOL: np.bool_,
       DT_UINT8: np.uint8}
_DT = {np.dtype(v): k for k, v in _NP.items()}


# ---- CRC32C (Castagnoli), masked as in tensorflow/core/lib/hash/crc32c.h ---- #
def _make_table():
  tab = []
  for i in range(256):
    c = i
    for _ in range(8):
      c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
    tab.append(c)
  return np.array(tab, dtype=np.uint32)


_TAB = _make_table()


def crc32c(data, crc=0):
  data = bytes(data)
  if len(data) >= 4096:                              # the library's host routine (csrc/error.cpp) for tensor payloads
    try:
      from seed_rl_amd import _lib
      return int(_lib.lib().seedhip_crc32c(data, len(data), crc))
    except Exception:                                # pylint: disable=broad-except
      pass
  c = (~crc) & 0xFFFFFFFF
  tab = _TAB
  for b in bytes(data):
    c = int(tab[(c ^ b) & 0xFF]) ^ (c >> 8)
  return (~c) & 0xFFFFFFFF


def _mask(crc):
  return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xFFFFFFFF


def _varint(n):
  out = bytearray()
  while True:
    b = n & 0x7F
    n >>= 7
    out.append(b | (0x80 if n else 0))
    if not n:
      return bytes(out)


def _read_varint(buf, pos):
  v, s = 0, 0
  while True:
    b = buf[pos]; pos += 1
    v |= (b & 0x7F) << s
    s += 7
    if not b & 0x80:
      return v, pos


def _snappy_uncompress(src):
  """Minimal Snappy block decoder (TF's table builder may compress blocks)."""
  n, pos = _read_varint(src, 0)
  out = bytearray()
  while pos < len(src):
    tag = src[pos]; pos += 1
    kind = tag & 3
    if kind == 0:
      ln = tag >> 2
      if ln >= 60:
        nb = ln - 59
        ln = int.from_bytes(src[pos:pos + nb], 'little'); pos += nb
      ln += 1
      out += src[pos:pos + ln]; pos += ln
      continue
    if kind == 1:
      ln, off = ((tag >> 2) & 7) + 4, ((tag >> 5) << 8) | src[pos]; pos += 1
    elif kind == 2:
      ln, off = (tag >> 2) + 1, int.from_bytes(src[pos:pos + 2], 'little'); pos += 2
    else:
      ln, off = (tag >> 2) + 1, int.from_bytes(src[pos:pos + 4], 'little'); pos += 4
    for _ in range(ln):
      out.append(out[-off])
  assert len(out) == n, 'corrupt snappy block'
  return bytes(out)


# ---- table (LevelDB format) ---- #
def _read_block(buf, offset, size):
  raw, typ = buf[offset:offset + size], buf[offset + size]
  stored = struct.unpack('<I', buf[offset + size + 1:offset + size + 5])[0]
  if _mask(crc32c(buf[offset:offset + size + 1])) != stored:
    raise ValueError('checkpoint index: block checksum mismatch at offset %d' % offset)
  if typ == 1:
    raw = _snappy_uncompress(raw)
  elif typ != 0:
    raise ValueError('checkpoint index: unknown block compression %d' % typ)
  return raw


def _block_entries(block):
  nrestarts = struct.unpack('<I', block[-4:])[0]
  end = len(block) - 4 - 4 * nrestarts
  pos, key = 0, b''
  while pos < end:
    shared, pos = _read_varint(block, pos)
    non_shared, pos = _read_varint(block, pos)
    vlen, pos = _read_varint(block, pos)
    key = key[:shared] + bytes(block[pos:pos + non_shared]); pos += non_shared
    yield key, bytes(block[pos:pos + vlen])
    pos += vlen


def _build_block(entries, restart_interval=16):
  out, restarts, last = bytearray(), [], b''
  for i, (k, v) in enumerate(entries):
    shared = 0
    if i % restart_interval == 0:
      restarts.append(len(out))
    else:
      while shared < min(len(k), len(last)) and k[shared] == last[shared]:
        shared += 1
    out += _varint(shared) + _varint(len(k) - shared) + _varint(len(v)) + k[shared:] + v
    last = k
  if not restarts:
    restarts = [0]
  for r in restarts:
    out += struct.pack('<I', r)
  out += struct.pack('<I', len(restarts))
  return bytes(out)


def _with_trailer(block):
  return block + b'\x00' + struct.pack('<I', _mask(crc32c(block + b'\x00')))


def read_index(path):
  """<prefix>.index -> OrderedDict key(bytes) -> value(bytes), in table order."""
  buf = open(path, 'rb').read()
  if len(buf) < 48 or struct.unpack('<Q', buf[-8:])[0] != _MAGIC:
    raise ValueError('%s is not a TensorFlow checkpoint index (bad magic)' % path)
  footer = buf[-48:]
  _, p = _read_varint(footer, 0)                    # metaindex handle
  _, p = _read_varint(footer, p)
  ioff, p = _read_varint(footer, p)
  isize, p = _read_varint(footer, p)
  out = collections.OrderedDict()
  for _, handle in _block_entries(_read_block(buf, ioff, isize)):
    off, q = _read_varint(handle, 0)
    size, _ = _read_varint(handle, q)
    for k, v in _block_entries(_read_block(buf, off, size)):
      out[k] = v
  return out


def read_checkpoint(prefix):
  """-> OrderedDict tensor name -> numpy array for every numeric tensor of the bundle <prefix>.index / .data-*."""
  index = read_index(prefix + '.index')
  header = BundleHeaderProto()
  header.ParseFromString(index.get(b'', b''))
  if header.endianness != 0:
    raise ValueError('big-endian checkpoints are not supported')
  shards = {}
  out = collections.OrderedDict()
  for k, v in index.items():
    if k == b'':
      continue
    e = BundleEntryProto()
    e.ParseFromString(v)
    if (e.dtype not in _NP and e.dtype != DT_STRING) or len(e.slices):
      continue                                       # variants, partitioned variables
    if e.shard_id not in shards:
      shards[e.shard_id] = open('%s.data-%05d-of-%05d' % (prefix, e.shard_id, max(header.num_shards, 1)), 'rb').read()
    raw = shards[e.shard_id][e.offset:e.offset + e.size]
    if e.dtype == DT_STRING:                          # (the object graph: a scalar string)
      n = 1
      for d in e.shape.dim:
        n *= int(d.size)
      strings, crc = _parse_string_tensor(raw, n)
      if e.crc32c and _mask(crc) != e.crc32c:
        raise ValueError('checkpoint data: checksum mismatch for %s' % k.decode())
      out[k.decode()] = strings[0] if not len(e.shape.dim) else np.array(strings, dtype=object).reshape(
          tuple(int(d.size) for d in e.shape.dim))
      continue
    if e.crc32c and _mask(crc32c(raw)) != e.crc32c:
      raise ValueError('checkpoint data: checksum mismatch for %s' % k.decode())
    shape = tuple(int(d.size) for d in e.shape.dim)
    out[k.decode()] = np.frombuffer(raw, dtype=_NP[e.dtype]).reshape(shape).copy()
  return out


def _string_tensor(strings):
  """tensor_bundle.cc WriteStringTensor: [varint64 length]* [4-byte masked crc32c of the lengths as uint64s] [bytes]*;
  the entry's checksum runs over the lengths (as uint64), the 4 checksum bytes and the string bytes."""
  out, crc = bytearray(), 0
  for b in strings:
    out += _varint(len(b))
    crc = crc32c(struct.pack('<Q', len(b)), crc)
  lc = struct.pack('<I', _mask(crc))
  out += lc
  crc = crc32c(lc, crc)
  for b in strings:
    out += b
    crc = crc32c(b, crc)
  return bytes(out), crc


def _parse_string_tensor(raw, n):
  lengths, p, crc = [], 0, 0
  for _ in range(n):
    v, p = _read_varint(raw, p)
    lengths.append(v)
    crc = crc32c(struct.pack('<Q', v), crc)
  if struct.unpack('<I', raw[p:p + 4])[0] != _mask(crc):
    raise ValueError('checkpoint data: string-length checksum mismatch')
  crc = crc32c(raw[p:p + 4], crc)
  p += 4
  strings = []
  for v in lengths:
    strings.append(bytes(raw[p:p + v]))
    p += v
  for b in strings:
    crc = crc32c(b, crc)
  return strings, crc


def object_graph(keys, optimizer_root='optimizer'):
  """TrackableObjectGraph for a set of tf.train.Checkpoint keys (those ending in /.ATTRIBUTES/VARIABLE_VALUE): node 0 is
  the Checkpoint object, every path component a child edge (`local_name`), a key's last node carries the
  VARIABLE_VALUE attribute; <variable path>/.OPTIMIZER_SLOT/<optimizer>/<slot> keys become nodes without a parent
  edge, referenced from the optimizer node's slot_variables (training/tracking/graph_view.py serialises them so)."""
  g = TrackableObjectGraph()
  g.nodes.add()
  ids = {(): 0}

  def node(path):
    path = tuple(path)
    if path not in ids:
      parent = node(path[:-1])
      ids[path] = len(g.nodes)
      g.nodes.add()
      ref = g.nodes[parent].children.add()
      ref.node_id, ref.local_name = ids[path], path[-1]
    return ids[path]

  def attribute(n, key, full_name):
    a = g.nodes[n].attributes.add()
    a.name, a.full_name, a.checkpoint_key = 'VARIABLE_VALUE', full_name, key

  slots = []
  for key in sorted(keys, key=lambda k: k.encode()):
    if not key.endswith(_SUFFIX):
      continue
    parts = key[:-len(_SUFFIX)].split('/')
    if '.OPTIMIZER_SLOT' in parts:
      slots.append((key, parts))
      continue
    attribute(node(parts), key, '/'.join(parts[1:]) or parts[0])
  for key, parts in slots:
    i = parts.index('.OPTIMIZER_SLOT')
    var, opt, slot = parts[:i], parts[i + 1:-1], parts[-1]
    n = len(g.nodes)
    g.nodes.add()
    attribute(n, key, '/'.join(var[1:]) + '/' + slot)
    ref = g.nodes[node(opt)].slot_variables.add()
    ref.original_variable_node_id, ref.slot_name, ref.slot_variable_node_id = node(var), slot, n
  return g


def walk_object_graph(graph_bytes):
  """{checkpoint key: path of local names from the root} of a serialized TrackableObjectGraph; slot variables as
  <variable path> + ['.OPTIMIZER_SLOT'] + <optimizer path> + [slot name] -- what restore() matches objects by."""
  g = TrackableObjectGraph()
  g.ParseFromString(graph_bytes)
  paths, out = {0: []}, {}
  todo = [0]
  while todo:
    n = todo.pop()
    for c in g.nodes[n].children:
      if c.node_id not in paths:
        paths[c.node_id] = paths[n] + [c.local_name]
        todo.append(c.node_id)
  for n, obj in enumerate(g.nodes):
    for s in obj.slot_variables:
      paths[s.slot_variable_node_id] = paths[s.original_variable_node_id] + ['.OPTIMIZER_SLOT'] + paths[n] + [s.slot_name]
  for n, obj in enumerate(g.nodes):
    for a in obj.attributes:
      out[a.checkpoint_key] = paths[n]
  return out


def write_checkpoint(prefix, tensors):
  """Writes <prefix>.index and <prefix>.data-00000-of-00001 for {name: array} (one shard, uncompressed blocks); a
  `bytes` value becomes a scalar string tensor."""
  data, entries = bytearray(), []
  header = BundleHeaderProto()
  header.num_shards = 1
  header.version.producer = 1
  entries.append((b'', header.SerializeToString()))
  for name in sorted(tensors, key=lambda n: n.encode()):
    if isinstance(tensors[name], (bytes, bytearray)):
      raw, crc = _string_tensor([bytes(tensors[name])])
      e = BundleEntryProto()
      e.dtype = DT_STRING
      e.shard_id, e.offset, e.size, e.crc32c = 0, len(data), len(raw), _mask(crc)
      data += raw
      entries.append((name.encode(), e.SerializeToString()))
      continue
    a = np.asarray(tensors[name])                   # (np.ascontiguousarray would turn scalars into shape (1,))
    a = a if a.flags.c_contiguous else a.copy()
    e = BundleEntryProto()
    e.dtype = _DT[a.dtype]
    for d in a.shape:
      e.shape.dim.add().size = int(d)
    raw = a.tobytes()
    e.shard_id, e.offset, e.size, e.crc32c = 0, len(data), len(raw), _mask(crc32c(raw))
    data += raw
    entries.append((name.encode(), e.SerializeToString()))
  with open(prefix + '.data-00000-of-00001', 'wb') as f:
    f.write(bytes(data))
  out = bytearray()
  index_entries = []
  for lo in range(0, len(entries), 64):              # one data block per 64 entries
    chunk = entries[lo:lo + 64]
    blk = _build_block(chunk)
    index_entries.append((chunk[-1][0], _varint(len(out)) + _varint(len(blk))))
    out += _with_trailer(blk)
  meta = _build_block([])
  meta_handle = _varint(len(out)) + _varint(len(meta))
  out += _with_trailer(meta)
  idx = _build_block(index_entries, restart_interval=1)
  idx_handle = _varint(len(out)) + _varint(len(idx))
  out += _with_trailer(idx)
  footer = meta_handle + idx_handle
  footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', _MAGIC)
  out += footer
  with open(prefix + '.index', 'wb') as f:
    f.write(bytes(out))


# ---- object-graph names of the reference agents ---- #
def _impala_deep_names(num_stacks):
  """dmlab/networks.py:26-89 / football/networks.py:27-96 attribute paths (without the agent prefix)."""
  m = collections.OrderedDict()
  for i in range(num_stacks):
    for w in ('kernel', 'bias'):
      m['stack%d/conv/%s' % (i, w)] = '_stacks/%d/_conv/%s' % (i, w)
      for b in range(2):
        for j in range(2):
          m['stack%d/res_%d/conv2d_%d/%s' % (i, b, j, w)] = '_stacks/%d/_res_convs%d/%d/%s' % (i, j, b, w)
  for ours, theirs in (('conv_to_linear', '_conv_to_linear'), ('policy_logits', '_policy_logits'), ('baseline', '_baseline')):
    for w in ('kernel', 'bias'):
      m['%s/%s' % (ours, w)] = '%s/%s' % (theirs, w)
  return m


def reference_variable_paths(agent):
  """{our reference-structured variable name: attribute path under the agent object} for the agents whose reference
  class layout is known (the path of a tf.train.Checkpoint key is '<root attr>/<path>/.ATTRIBUTES/VARIABLE_VALUE')."""
  from seed_rl_amd import networks
  if isinstance(agent, networks.GFootball):
    return _impala_deep_names(4)
  if isinstance(agent, networks.ImpalaDeep):
    m = _impala_deep_names(len(agent._channels))      # pylint: disable=protected-access
    for w in ('kernel', 'recurrent_kernel', 'bias'):
      m['core/' + w] = '_core/' + w
    return m
  if isinstance(agent, networks.DuelingLSTMDQNNet):   # atari/networks.py:232-252 (Sequential children by index)
    m = collections.OrderedDict()
    for i in range(3):
      for w in ('kernel', 'bias'):
        m['body/conv%d/%s' % (i, w)] = '_body/layer_with_weights-%d/%s' % (i, w)
    for w in ('kernel', 'bias'):
      m['body/fc/' + w] = '_body/layer_with_weights-3/' + w
      m['value/hidden/' + w] = '_value/layer_with_weights-0/' + w
      m['value/head/' + w] = '_value/layer_with_weights-1/' + w
      m['advantage/hidden/' + w] = '_advantage/layer_with_weights-0/' + w
    m['advantage/head/kernel'] = '_advantage/layer_with_weights-1/kernel'
    for w in ('kernel', 'recurrent_kernel', 'bias'):
      m['core/' + w] = '_core/' + w
    return m
  if isinstance(agent, networks.MLPandLSTM):          # agents/vtrace/networks.py:41-52
    m = collections.OrderedDict()
    for i in range(len(agent._mlp)):                  # pylint: disable=protected-access
      for w in ('kernel', 'bias'):
        m['mlp/dense_%d/%s' % (i, w)] = '_mlp/layer_with_weights-%d/%s' % (i, w)
    for l in range(len(agent._lstm)):                 # pylint: disable=protected-access
      for w in ('kernel', 'recurrent_kernel', 'bias'):
        m['core/cell_%d/%s' % (l, w)] = '_core/cells/%d/%s' % (l, w)
    for ours, theirs in (('policy_logits', '_policy_logits'), ('baseline', '_baseline')):
      for w in ('kernel', 'bias'):
        m['%s/%s' % (ours, w)] = '%s/%s' % (theirs, w)
    return m
  raise ValueError('no reference class layout for %s (AtariShallow has no reference counterpart: SURVEY.md D1)'
                   % type(agent).__name__)


def restore_agent(prefix, agent, root='agent', optimizer=None, optimizer_root='optimizer'):
  """Loads a tf.train.Checkpoint written by the reference learner: the agent's variables (root 'agent' /
  'target_agent'), and with `optimizer` Adam's iteration count and first / second moment slots."""
  import torch
  tensors = read_checkpoint(prefix)
  paths = reference_variable_paths(agent)
  values = {}
  ec_key = '%s/entropy_cost_param%s' % (root, _SUFFIX)
  if ec_key in tensors:                               # whether or not a Learner has attached the parameter yet
    values['entropy_cost_param'] = np.asarray(tensors[ec_key]).reshape(1)
  for name, _, _ in agent._ref_spec:                  # pylint: disable=protected-access
    if name == 'entropy_cost_param':
      continue
    key = '%s/%s%s' % (root, paths[name], _SUFFIX)
    if key not in tensors:
      raise KeyError('checkpoint %s has no variable %s' % (prefix, key))
    values[name] = tensors[key]
  agent.load_reference_params(values)
  if optimizer is None:
    return sorted(values)
  from seed_rl_amd import checkpoint as ckpt
  m = torch.zeros_like(agent.flat.params)
  v = torch.zeros_like(agent.flat.params)
  for slot, buf in (('m', m), ('v', v)):
    for name, view in ckpt._ref_views(agent, buf).items():   # pylint: disable=protected-access
      path = paths.get(name, name)
      key = '%s/%s/.OPTIMIZER_SLOT/%s/%s%s' % (root, path, optimizer_root, slot, _SUFFIX)
      if key in tensors:
        view.copy_(torch.as_tensor(tensors[key]).to(buf.device).reshape(view.shape))
  it = tensors.get('%s/iter%s' % (optimizer_root, _SUFFIX))
  optimizer.load_state_dict(dict(iterations=int(it) if it is not None else 0, m=m, v=v))
  return sorted(values)


def save_agent(prefix, agent, root='agent', optimizer=None, optimizer_root='optimizer', extra=None):
  """Writes the agent (and Adam state) under the reference's tf.train.Checkpoint keys."""
  from seed_rl_amd import checkpoint as ckpt
  paths = reference_variable_paths(agent)
  tensors = dict(extra or {})
  for name, view in agent.trainable_variables:
    a = view.detach().cpu().numpy()
    if name == 'entropy_cost_param':
      a = a.reshape(())                               # a scalar variable in the reference (learner.py:225-234)
    tensors['%s/%s%s' % (root, paths.get(name, name), _SUFFIX)] = a
  if optimizer is not None:
    sd = optimizer.state_dict()
    tensors['%s/iter%s' % (optimizer_root, _SUFFIX)] = np.asarray(sd['iterations'], np.int64)
    if sd['m'] is not None:
      for slot, buf in (('m', sd['m']), ('v', sd['v'])):
        for name, view in ckpt._ref_views(agent, buf).items():   # pylint: disable=protected-access
          key = '%s/%s/.OPTIMIZER_SLOT/%s/%s%s' % (root, paths.get(name, name), optimizer_root, slot, _SUFFIX)
          a = view.detach().cpu().numpy()
          tensors[key] = a.reshape(()) if name == 'entropy_cost_param' else a
  if 'save_counter' + _SUFFIX not in tensors:         # tf.train.Checkpoint.save() counts its calls there and names the
    # file `<dir>/ckpt-<save_counter>` (learner.py:469-475): take it back -- only from names of exactly that form (a
    # prefix such as `run-3` or `model-2024` is not a CheckpointManager number: ADVICE r5)
    m = re.fullmatch(r'ckpt-(\d+)', os.path.basename(prefix))
    tensors['save_counter' + _SUFFIX] = np.asarray(int(m.group(1)) if m else 1, np.int64)
  tensors[OBJECT_GRAPH_KEY] = object_graph(tensors, optimizer_root).SerializeToString()
  write_checkpoint(prefix, tensors)
  return sorted(tensors)

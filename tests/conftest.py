import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run via gpurun)')


@pytest.fixture(scope='session')
def device():
  import torch
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  return torch.device('cuda:0')


def pytest_collection_modifyitems(config, items):
  """A per-test wall-clock limit (pytest-timeout, when installed): a test that waits for something that never comes --
  closed-loop actors whose last batch cannot fill, a peer process that died -- fails with the stacks of all threads
  instead of sitting there until the outer limit kills the whole run without a word."""
  if not config.pluginmanager.hasplugin('timeout'):
    return
  for item in items:
    if item.get_closest_marker('timeout') is None:
      item.add_marker(pytest.mark.timeout(900 if item.get_closest_marker('gpu') else 300))

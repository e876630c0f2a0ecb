import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import test_gpu_undulator as t
for tag in t.CASES:
    g = t.load('tests/golden', tag)
    Is, Ip = t.run_dev(g)
    print(tag, t.rel(Is, g['Is']), t.rel(Ip, g['Ip']))

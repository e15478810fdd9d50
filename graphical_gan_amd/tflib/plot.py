"""tflib/plot.py shim: same tick / plot / flush names; flush prints the means since the last flush and
appends them to the logfile (the per-metric matplotlib JPGs of the reference are not produced)."""
import collections

import numpy as np

_since_beginning = collections.defaultdict(lambda: {})
_since_last_flush = collections.defaultdict(lambda: {})
_iter = [0]


def tick():
    _iter[0] += 1


def plot(name, value):
    _since_last_flush[name][_iter[0]] = float(value)


def flush(outf=None, logfile=None):
    prints = []
    for name, vals in _since_last_flush.items():
        prints.append("{}\t{}".format(name, np.mean(list(vals.values()))))
        _since_beginning[name].update(vals)
    line = "iter {}\t{}".format(_iter[0], "\t".join(prints))
    print(line)
    if logfile:
        with open(logfile, 'a') as f:
            f.write(line + "\n")
    _since_last_flush.clear()

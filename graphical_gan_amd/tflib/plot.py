"""tflib/plot.py (/root/reference/tflib/plot.py:1-40): same tick / plot / flush names and signatures.  flush prints the means since
the last flush, appends them to `logfile` and saves one <metric>.jpg per metric under `outf` (matplotlib, Agg backend) as the
reference does; `outf` / `logfile` may be None (no files: the driver loop without an output directory)."""
import collections
import os

import numpy as np

_since_beginning = collections.defaultdict(lambda: {})
_since_last_flush = collections.defaultdict(lambda: {})
_iter = [0]


def tick():
    _iter[0] += 1


def plot(name, value):
    _since_last_flush[name][_iter[0]] = float(value)


def flush(outf, logfile):
    prints = []
    plt = None
    if outf:
        try:
            import matplotlib
            matplotlib.use('Agg')
            import matplotlib.pyplot as plt
        except ImportError:                    # the curves are a convenience; the log line is the record
            plt = None
    for name, vals in _since_last_flush.items():
        prints.append("{}\t{}".format(name, np.mean(list(vals.values()))))
        _since_beginning[name].update(vals)
        if plt is not None:
            xs = sorted(_since_beginning[name])
            plt.clf()
            plt.plot(xs, [_since_beginning[name][x] for x in xs])
            plt.xlabel('iteration')
            plt.ylabel(name)
            plt.savefig(os.path.join(outf, name.replace(' ', '_') + '.jpg'))
    line = "iter {}\t{}".format(_iter[0], "\t".join(prints))
    print(line)
    if logfile:
        with open(logfile, 'a') as f:
            f.write(line + "\n")
    _since_last_flush.clear()

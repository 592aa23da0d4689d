"""Logging setup (cvangysel.logging_utils call sites: bin/train.py:70-75,
bin/query.py:47-50)."""
import logging
import sys


def configure_logging(args, output_path=None):
    level = getattr(logging, str(getattr(args, 'loglevel', 'INFO')).upper(), None)
    if not isinstance(level, int):
        raise IOError('Invalid log level: {0}'.format(args.loglevel))
    handlers = [logging.StreamHandler(sys.stderr)]
    if output_path:
        handlers.append(logging.FileHandler(output_path))
    logging.basicConfig(level=level, handlers=handlers, force=True,
                        format='%(asctime)s [%(threadName)s] [%(levelname)s]  %(message)s')
    logging.info('Arguments: %s', args)


def log_module_info(*modules):
    for module in modules:
        logging.info('%s version: %s (%s)', getattr(module, '__name__', module),
                     getattr(module, '__version__', 'n/a'),
                     getattr(module, '__file__', 'n/a'))

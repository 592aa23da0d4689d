"""Minimal in-repo equivalents of the un-vendored ``cvangysel-common`` helpers the
two hot-path CLIs call: argparse validators, logging setup, TREC
topic / run I/O, the word2vec-binary loader.  The upstream sources are an empty
git submodule in the reference (.gitmodules:1-3); behaviour follows the call
sites in bin/train.py and bin/query.py."""

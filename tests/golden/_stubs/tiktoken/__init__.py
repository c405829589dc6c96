"""Test-tooling stub: never used on the retrieval path."""


def get_encoding(name):
    raise RuntimeError("tiktoken stub")


def encoding_for_model(name):
    raise RuntimeError("tiktoken stub")

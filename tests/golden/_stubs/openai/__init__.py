"""Test-tooling stub: names only; no network client is ever constructed."""


class OpenAI:
    def __init__(self, *a, **k):
        pass


class AsyncOpenAI(OpenAI):
    pass

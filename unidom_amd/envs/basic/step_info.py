"""step_diff's info dict, shared by the cloth and MPM env bases."""


class _StepInfo(dict):
    """step_diff's info dict; entries registered as thunks are computed on first access."""

    def __init__(self, eager, **lazy):
        super().__init__(eager)
        self._lazy = lazy

    def __missing__(self, key):
        if key in self._lazy:
            self[key] = self._lazy.pop(key)()
            return dict.__getitem__(self, key)
        raise KeyError(key)

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._lazy

    def keys(self):
        return list(dict.keys(self)) + list(self._lazy.keys())

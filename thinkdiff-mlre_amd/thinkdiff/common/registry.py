"""Name -> class registry with the reference's API surface (thinkdiff/common/registry.py:9-329):
`registry.register_model("name")` decorators, `registry.get_model_class(name) -> cls | None`, paths."""


class Registry:
    _KINDS = ("builder", "task", "model", "processor", "lr_scheduler", "runner")

    def __init__(self):
        self.mapping = {f"{k}_name_mapping": {} for k in self._KINDS}
        self.mapping["paths"] = {}
        self.mapping["state"] = {}
        for kind in self._KINDS:
            setattr(self, f"register_{kind}", self._make_register(kind))
            setattr(self, f"get_{kind}_class", self._make_get(kind))
            setattr(self, f"list_{kind}s", self._make_list(kind))

    def _make_register(self, kind):
        table = self.mapping[f"{kind}_name_mapping"]

        def register(name):
            def wrap(cls):
                if name in table and table[name] is not cls:
                    raise KeyError(f"Name '{name}' already registered for {table[name]}.")
                table[name] = cls
                return cls
            return wrap
        return register

    def _make_get(self, kind):
        table = self.mapping[f"{kind}_name_mapping"]
        return lambda name: table.get(name, None)

    def _make_list(self, kind):
        table = self.mapping[f"{kind}_name_mapping"]
        return lambda: sorted(table.keys())

    def register_path(self, name, path):
        assert isinstance(path, str), "All path must be str."
        if name in self.mapping["paths"]:
            raise KeyError(f"Name '{name}' already registered.")
        self.mapping["paths"][name] = path

    def get_path(self, name):
        return self.mapping["paths"].get(name, None)

    def register(self, name, obj):
        cur = self.mapping["state"]
        parts = name.split(".")
        for p in parts[:-1]:
            cur = cur.setdefault(p, {})
        cur[parts[-1]] = obj

    def get(self, name, default=None, no_warning=False):
        cur = self.mapping["state"]
        for p in name.split("."):
            if not isinstance(cur, dict) or p not in cur:
                return default
            cur = cur[p]
        return cur

    def unregister(self, name):
        return self.mapping["state"].pop(name, None)


registry = Registry()

"""see the module of the same name in this package."""

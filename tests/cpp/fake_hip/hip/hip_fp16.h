// (fake: nothing in the host shim uses half types)

"""MI355X-native engine behind the reference's lib/models builder surface."""

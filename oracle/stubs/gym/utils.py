seeding = None
